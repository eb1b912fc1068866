"""CPU tests of the config entry point (configs/*.yaml + unirestore_amd/cli.py), the synthetic data module and the
caller-side metrics.  Key schema follows the reference's LightningCLI files (/root/reference/configs/val.yaml:6-12,47-67)."""
import glob
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = sorted(glob.glob(os.path.join(ROOT, "configs", "*.yaml")))


def test_five_baseline_configs_exist():
    names = [os.path.basename(c) for c in CONFIGS]
    assert len(names) == 5, names


@pytest.mark.parametrize("path", CONFIGS, ids=[os.path.basename(c) for c in CONFIGS])
def test_config_resolves(path):
    from unirestore_amd import cli
    cfg = cli.load_config(path)
    assert set(cfg) >= {"seed_everything", "trainer", "model", "data"}
    mk = cfg["model"]["init_args"]["model_kwargs"]
    assert set(mk) == {"frenc", "cnet", "tedit"} and mk["frenc"]["type"] == "CFRM" and mk["cnet"]["type"] in ("scedit", "spade")
    assert mk["tedit"]["type"] == "TFA" and mk["tedit"]["task"] == ["ir", "cls", "seg"]
    r = cli.resolve(cfg)
    assert r["dtype"] in ("bf16", "fp16") and r["devices"] in (1, 4, 8)
    assert r["model_kwargs"]["cnet"]["num_inference_steps"] in (4, 20, 50)
    from unirestore_amd.data import SyntheticImages
    ds = SyntheticImages(**r["data_args"])
    assert ds.batch_size % r["devices"] == 0 and ds.task in ("ir", "seg")


def test_configs_cover_baseline_json():
    from unirestore_amd import cli
    got = {}
    for c in CONFIGS:
        r = cli.resolve(cli.load_config(c))
        d = r["data_args"]
        got[os.path.basename(c)] = (tuple(d["resolution"]), d["batch_size"], r["model_kwargs"]["cnet"]["num_inference_steps"], r["dtype"],
                                    r["devices"], d["task"])
    assert got["val_pir_256_4step.yaml"] == ((256, 256), 1, 4, "bf16", 1, "ir")
    assert got["val_pir_512_b8_20step_bf16.yaml"] == ((512, 512), 8, 20, "bf16", 1, "ir")
    assert got["val_pir_512_b64_20step_8gpu.yaml"] == ((512, 512), 64, 20, "bf16", 8, "ir")
    assert got["val_tir_seg_1024_20step_4gpu.yaml"] == ((1024, 1024), 4, 20, "bf16", 4, "seg")
    assert got["val_mixed_512_50step_fp16_8gpu.yaml"] == ((512, 512), 64, 50, "fp16", 8, "ir")


def test_reference_style_config_and_overrides(tmp_path):
    """A file in the reference's own layout (class paths, $placeholder$ checkpoints, data.DatasetEngine, precision 32)."""
    from unirestore_amd import cli
    p = tmp_path / "val.yaml"
    p.write_text("""
seed_everything: 42
trainer: {accelerator: gpu, devices: [0, 1], precision: 32}
model:
  class_path: core.engine_unifie.LitUniFIEIR
  init_args:
    save_image: False
    eval_mode: ALL
    need_crop: True
    model_kwargs:
      frenc: {train: false, ckpt_path: $path_to_stage1_ckpt$, type: CFRM}
      cnet: {train: false, ckpt_path: $path_to_stage1_ckpt$, type: scedit, num_inference_steps: 1}
      tedit: {ckpt_path: $path_to_stage2_ckpt$, type: TFA, prompt_len: 1, task: ["ir", "cls", "seg"], train: false}
data:
  class_path: data.DatasetEngine
  init_args: {task: mtl, train: {type: all, resolution: 512, batch_size: 1}, val: {type: val, val_list: [], batch_size: 1}}
""")
    with pytest.raises(ValueError, match="allow-16bit"):                 # precision 32 is refused, never silently narrowed
        cli.resolve(cli.load_config(str(p)))
    with pytest.warns(UserWarning, match="fp16"):
        r = cli.resolve(cli.load_config(str(p), ["model.init_args.model_kwargs.cnet.num_inference_steps=4"]), allow_16bit=True)
    assert r["dtype"] == "fp16"
    assert r["model_kwargs"]["cnet"]["num_inference_steps"] == 4 and r["model_kwargs"]["cnet"]["ckpt_path"] is None
    assert r["devices"] == 2 and r["data_args"]["resolution"] == [512, 512] and r["caller_args"]["need_crop"] is True
    with pytest.raises(ValueError):
        cli.resolve(dict(trainer=dict(accelerator="cpu"), model=cli.load_config(str(p))["model"]))       # no CPU fallback
    with pytest.raises(KeyError):
        cli.resolve(dict(model=dict(class_path="core.engine_seg.Something", init_args=dict(model_kwargs={}))))
    with pytest.raises(ValueError):
        cli.resolve(dict(trainer=dict(precision="64"), model=cli.load_config(str(p))["model"]))


def test_synthetic_data_is_world_size_independent():
    from unirestore_amd.data import SyntheticImages, degrade
    ds = SyntheticImages(resolution=[8, 12], batch_size=6, num_batches=2, degradations=["noise", "haze", "lowlight"])
    whole = list(ds.batches(0, 1))
    parts = [list(ds.batches(r, 4)) for r in range(4)]
    for b in range(2):
        lq = torch.cat([parts[r][b][0] for r in range(4)])
        assert torch.equal(lq, whole[b][0]) and whole[b][0].shape == (6, 3, 8, 12) and whole[b][4] == "ir"
        assert sum(len(parts[r][b][3]) for r in range(4)) == 6
    hq = whole[0][1]
    assert torch.equal(whole[0][0][1], degrade(hq[1], "haze")) and torch.equal(whole[0][0][2], degrade(hq[2], "lowlight"))
    with pytest.raises(ValueError):
        SyntheticImages(degradations=["fog"])


def test_ssim_and_psnr_match_the_textbook_definitions():
    """SSIM per scikit-image's defaults (uniform 7x7 window, sample covariance, cropped borders), restated with scipy."""
    from scipy.ndimage import uniform_filter
    from unirestore_amd import runner
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2, 3, 24, 31, generator=g)
    y = (x + 0.1 * torch.randn(x.shape, generator=g)).clamp(0, 1)

    def sk_ssim(a, b, win=7, R=1.0):
        vals = []
        for n in range(a.shape[0]):
            for c in range(a.shape[1]):
                im1, im2 = a[n, c].double().numpy(), b[n, c].double().numpy()
                NP = win * win
                cov_norm = NP / (NP - 1)
                ux, uy = uniform_filter(im1, win), uniform_filter(im2, win)
                uxx, uyy, uxy = uniform_filter(im1 * im1, win), uniform_filter(im2 * im2, win), uniform_filter(im1 * im2, win)
                vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
                C1, C2 = (0.01 * R) ** 2, (0.03 * R) ** 2
                S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
                pad = (win - 1) // 2
                vals.append(S[pad:-pad, pad:-pad].mean())
        return float(np.mean(vals))

    assert abs(runner.ssim(y, x) - sk_ssim(y, x)) < 1e-9
    assert abs(runner.ssim(x, x) - 1.0) < 1e-12
    # PSNR: skimage's per-image value, averaged over the images (the reference's SKPSNR sums per-sample PSNRs and divides by the
    # image count, eval_image_restoration.py:266-278) - NOT the PSNR of the batch-mean MSE; the two images differ on purpose
    y2 = y.clone()
    y2[1] = (x[1] + 0.3 * torch.randn(x[1].shape, generator=g)).clamp(0, 1)
    per = [10 * np.log10(1.0 / float(((y2[n].double() - x[n].double()) ** 2).mean())) for n in range(2)]
    assert abs(per[0] - per[1]) > 3.0
    assert abs(runner.psnr(y2, x) - float(np.mean(per))) < 1e-9
    assert torch.allclose(runner.psnr_per_image(y2, x), torch.tensor(per, dtype=torch.float64), atol=1e-9)
    batch_mse = float(((y2.double() - x.double()) ** 2).mean())
    assert abs(runner.psnr(y2, x) - 10 * np.log10(1.0 / batch_mse)) > 0.5          # the old (wrong) definition differs


def test_lit_metrics_accumulate_per_image_and_ir_task():
    """LitUniFIE's metric states: sum of per-image PSNR / image count; the IR evaluator restores with the 'ir' prompt whatever tag
    the batch carries (eval_image_restoration.py:70)."""
    from unirestore_amd import runner

    class Fake:
        calls = []

        def forward(self, imgs, task, quantize=False):
            Fake.calls.append(task)
            return imgs * 0.9
    lit = runner.LitUniFIE(dict(tedit=dict(task=["ir", "seg"])), model=Fake(), need_crop=False)
    g = torch.Generator().manual_seed(1)
    hq = torch.rand(3, 3, 16, 16, generator=g)
    lq = hq.clone()
    lq[2] = lq[2] * 0.5
    lit.validation_step((lq, hq, None, ["a", "b", "c"], "seg"))
    assert Fake.calls == ["ir"]
    m = lit.metrics()
    per = runner.psnr_per_image(lq * 0.9, hq)
    assert m["images"] == 3 and abs(m["val_lq/psnr"] - float(per.mean())) < 1e-9
