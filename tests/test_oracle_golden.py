"""Pin the oracle against vectors produced by the reference's own classes (tools/gen_golden.py)."""
import numpy as np
import pytest
import torch

from golden_util import golden_names, load_golden, rel_l2
from oracle import adapters, schedule
from oracle.blocks import AutoencoderKL, UNet2DConditionModel
from oracle.model import Controller, center_crop_box, resize_pad_plan, stablesr_config

TOL = 2e-6   # fp32 vs fp32, same op order up to reassociation


def _tfa_shape(w):
    cs = w["t_gate1.weight"].shape[1]
    co = w["conv_out.weight"].shape[0]
    t = w["out_gate.0.weight"].shape[1] // cs
    return co, cs, t, "prompt_trans.0.weight" not in w


@pytest.mark.parametrize("name", golden_names("csce"))
def test_csce(name):
    w, i, o = load_golden(name)
    m = adapters.CSCEAdapter(w["proj.weight"].shape[0], w["tuner.0.weight"].shape[0], w["proj.weight"].shape[1]).eval()
    m.load_state_dict(w)
    assert rel_l2(m(i["x"], i["condition"]), o["y"]) < TOL


@pytest.mark.parametrize("name", golden_names("tfa"))
def test_tfa(name):
    w, i, o = load_golden(name)
    m = adapters.TaskFeatureAdapter(*_tfa_shape(w)).eval()
    m.load_state_dict(w)
    x, c = m(i["x"], i["skip"], i["condition"])
    assert rel_l2(x, o["x"]) < TOL
    if "condition" in o:
        assert rel_l2(c, o["condition"]) < TOL
    else:
        assert c is None


@pytest.mark.parametrize("name", golden_names("cfrm"))
def test_cfrm(name):
    w, i, o = load_golden(name)
    c = w["0.conv1.weight"].shape[1]
    n = max(int(k.split(".")[0]) for k in w)
    m = adapters.cfrm_blocks((c,), (n,))[0].eval()
    m.load_state_dict(w)
    assert rel_l2(m(i["x"]), o["y"]) < TOL


def test_spade():
    w, i, o = load_golden("spade_0")
    m = adapters.SPADE(64, 32).eval()
    m.load_state_dict(w)
    assert rel_l2(m(i["x"], i["segmap"]), o["y"]) < TOL


def test_schedule_kats():
    """integer schedule bit-exact, fp32 alphas_cumprod bit-exact."""
    ac = schedule.alphas_cumprod()
    kat = {0: 0.9991499781608582, 49: 0.9526252746582031, 249: 0.6754320859909058,
           499: 0.27766942977905273, 749: 0.05662344768643379, 999: 0.00466009508818388}
    for t, v in kat.items():
        assert ac[t].item() == v
    assert schedule.ddim_timesteps(1).tolist() == [999]
    assert schedule.ddim_timesteps(4).tolist() == [999, 749, 499, 249]        # == train_timesteps, unifie.py:67
    assert schedule.ddim_timesteps(20).tolist() == list(range(999, 0, -50))
    assert schedule.ddim_timesteps(50).tolist() == list(range(999, 0, -20))
    assert schedule.ddim_timesteps(20).dtype == np.int64


def test_ddim_closed_form():
    g = torch.Generator().manual_seed(0)
    x, e = torch.randn(2, 4, 8, 8, generator=g), torch.randn(2, 4, 8, 8, generator=g)
    ac = schedule.alphas_cumprod().double()
    y = schedule.ddim_step(e, 999, x, 4)
    x0 = (x.double() - (1 - ac[999]).sqrt() * e.double()) / ac[999].sqrt()
    ref = ac[749].sqrt() * x0 + (1 - ac[749]).sqrt() * e.double()
    assert rel_l2(y, ref) < 1e-6
    # last step: t_prev < 0 -> alpha_cumprod[0] (set_alpha_to_one=False)
    y = schedule.ddim_step(e, 249, x, 4)
    x0 = (x.double() - (1 - ac[249]).sqrt() * e.double()) / ac[249].sqrt()
    assert rel_l2(y, ac[0].sqrt() * x0 + (1 - ac[0]).sqrt() * e.double()) < 1e-6
    n = schedule.add_noise(x, e, torch.tensor([999, 249]))
    assert rel_l2(n[1], ac[249].sqrt() * x[1].double() + (1 - ac[249]).sqrt() * e[1].double()) < 1e-6


def test_param_counts_full_size():
    """Published sizes: UNet 865 910 724, VAE 83 653 863; Controller 52 494 080."""
    n = lambda m: sum(p.numel() for p in m.parameters())
    with torch.device("meta"):
        assert n(UNet2DConditionModel()) == 865_910_724
        assert n(AutoencoderKL()) == 83_653_863
        assert n(Controller(**stablesr_config)) == 52_494_080
        assert [n(adapters.CSCEAdapter(c, c, 256)) for c in (320, 640, 1280)] == [287_680, 984_960, 3_608_320]
        assert [n(adapters.TaskFeatureAdapter(512, s, 1, l)) for s, l in ((512, False), (256, False), (128, True))] == \
            [15_602_944, 4_164_480, 1_263_232]
        assert [n(b) for b in adapters.cfrm_blocks()] == [543_632, 2_135_824, 23_281_168]


def test_resize_pad_crop_integer_kats():
    """unifie.py:121-134 and eval_image_restoration.py:113-136."""
    assert resize_pad_plan(256, 256) == (512, 512, 0, 0)
    assert resize_pad_plan(300, 500) == (512, 853, 0, 43)
    assert resize_pad_plan(512, 512) == (512, 512, 0, 0)
    assert resize_pad_plan(720, 1280) == (720, 1280, 48, 0)
    assert center_crop_box(720, 1280) == (104, 616, 384, 896)
    assert center_crop_box(300, 500) == (0, 300, 0, 500)


def test_caller_side_crop_arithmetic():
    """crop_tensor window + resize/pad plan KATs from SURVEY.md 8(c): the integer arithmetic either side of the path."""
    import torch
    from oracle.model import center_crop_box, resize_pad_plan
    from unirestore_amd import runner
    from unirestore_amd.modules.model import resize_pad_plan as rp_hip
    for h, w in [(720, 1280), (512, 512), (300, 500), (513, 1023), (100, 2000), (511, 511)]:
        assert runner.crop_box(h, w) == center_crop_box(h, w)
        assert rp_hip(h, w) == resize_pad_plan(h, w)
    assert runner.crop_box(720, 1280) == (104, 616, 384, 896)
    assert runner.crop_tensor(torch.zeros(1, 3, 720, 1280)).shape == (1, 3, 512, 512)
    assert runner.crop_tensor(torch.zeros(3, 513, 300)).shape == (3, 512, 300)
    assert resize_pad_plan(256, 256) == (512, 512, 0, 0)
    assert resize_pad_plan(300, 500) == (512, 853, 0, 43)
    assert resize_pad_plan(720, 1280) == (720, 1280, 48, 0)
