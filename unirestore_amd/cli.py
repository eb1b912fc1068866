"""Config entry point: `python -m unirestore_amd.cli validate --config configs/<file>.yaml [--set a.b.c=value ...]`.

Resolves a LightningCLI-style YAML (the key schema of the reference's configs/*.yaml: `seed_everything`, `trainer.{accelerator,
devices,precision}`, `model.class_path` + `init_args.model_kwargs.{frenc,cnet,tedit}`, `data.class_path` + `init_args`;
reference src/main.py:17-28, configs/val.yaml:47-67, src/core/engine_unifie.py:29-42) into the MI355X path: model
(`checkpoint.build_from_config`), caller (`runner.LitUniFIE`), synthetic data (`data.SyntheticImages`), and runs the
validation loop.  One process per GPU: under `torch.distributed.run` the global batch is sharded over the ranks, the
weights are broadcast from rank 0 and the restored shards all-gathered over RCCL.  There is no CPU path: `accelerator: cpu`
is an error, not a fallback.  Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import yaml

# class paths the reference's files name -> the Lightning-free callers of this package
MODEL_CLASSES = {
    "core.engine_unifie.LitUniFIEIR": "unirestore_amd.runner.LitUniFIE",
    "core.engine_unifie.LitUniFIEMTL": "unirestore_amd.runner.LitUniFIE",
    "core.engine_unifie.LitUniFIE": "unirestore_amd.runner.LitUniFIE",
    "unirestore_amd.runner.LitUniFIE": "unirestore_amd.runner.LitUniFIE",
}
DATA_CLASSES = {"unirestore_amd.data.SyntheticImages": "unirestore_amd.data.SyntheticImages",
                "data.DatasetEngine": "unirestore_amd.data.SyntheticImages"}      # datasets are out of scope: synthetic stand-in
PRECISIONS = {"bf16-mixed": "bf16", "bf16": "bf16", "bf16-true": "bf16", "16-mixed": "fp16", "16": "fp16", "16-true": "fp16",
              "fp16": "fp16"}
# precision: 32 (the reference's shipped val.yaml default) has no fp32 matrix path here.  It is REFUSED unless the caller opts into
# the closest 16-bit type (fp16 storage / MFMA operands, fp32 accumulation, statistics, softmax and DDIM state: outputs within
# 1e-3 rel-L2 of the fp32 reference) with --allow-16bit / allow_16bit=True - never silently.
PRECISIONS_32 = ("32", "32-true")


def load_config(path: str, overrides=()) -> dict:
    with open(path) as f:
        cfg = yaml.safe_load(f)
    for ov in overrides:
        key, _, val = ov.partition("=")
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = yaml.safe_load(val)
    return cfg


def resolve(cfg: dict, allow_16bit: bool = False) -> dict:
    """Validate a config and reduce it to what the runner needs (raises ValueError / KeyError on unknown pieces)."""
    tr, mo, da = cfg.get("trainer", {}), cfg["model"], cfg.get("data", {})
    if str(tr.get("accelerator", "gpu")) not in ("gpu", "cuda", "auto"):
        raise ValueError(f"trainer.accelerator={tr.get('accelerator')!r}: the MI355X path has no CPU fallback (use accelerator: gpu)")
    prec = str(tr.get("precision", "bf16-mixed"))
    if prec in PRECISIONS_32:
        allow_16bit = allow_16bit or os.environ.get("UR_ALLOW_16BIT", "0") == "1"      # opt-in without editing the command line
        if not allow_16bit:
            raise ValueError(f"trainer.precision={prec!r}: this path computes with 16-bit MFMA operands (fp32 accumulation); pass "
                             "--allow-16bit (allow_16bit=True, or UR_ALLOW_16BIT=1 in the environment) to run the config in fp16, or set precision to bf16-mixed / 16-mixed")
        import warnings
        warnings.warn(f"trainer.precision={prec!r} runs as fp16 storage + fp32 accumulation (no fp32 matrix path on this backend)")
        dtype = "fp16"
    elif prec in PRECISIONS:
        dtype = PRECISIONS[prec]
    else:
        raise ValueError(f"trainer.precision={prec!r} not supported: choose from {sorted(PRECISIONS) + list(PRECISIONS_32)}")
    if mo["class_path"] not in MODEL_CLASSES:
        raise KeyError(f"model.class_path {mo['class_path']!r} is not a caller of the restoration path: {sorted(MODEL_CLASSES)}")
    init = dict(mo.get("init_args", {}))
    mk = init.pop("model_kwargs")
    for k in ("frenc", "cnet", "tedit"):
        if mk.get(k) and str(mk[k].get("ckpt_path", "")).startswith("$"):      # "$path_to_stage1_ckpt$" placeholders of val.yaml
            mk[k] = dict(mk[k], ckpt_path=None)
    dcp = da.get("class_path", "unirestore_amd.data.SyntheticImages")
    if dcp not in DATA_CLASSES:
        raise KeyError(f"data.class_path {dcp!r} unknown: {sorted(DATA_CLASSES)}")
    dargs = dict(da.get("init_args", {}))
    if dcp == "data.DatasetEngine":                       # the reference's key layout -> the synthetic stand-in's arguments
        val, train = dargs.get("val", {}), dargs.get("train", {})
        res = train.get("resolution", 512)
        dargs = dict(task={"mtl": "ir"}.get(dargs.get("task", "ir"), dargs.get("task", "ir")), resolution=[res, res],
                     batch_size=val.get("batch_size", 1), num_batches=4)
    devices = tr.get("devices", 1)
    n_dev = len(devices) if isinstance(devices, (list, tuple)) else (int(devices) if str(devices).isdigit() else 1)
    return dict(seed=cfg.get("seed_everything", 42), dtype=dtype, devices=n_dev, model_kwargs=mk,
                caller_args={k: init[k] for k in ("save_image", "eval_mode", "need_crop") if k in init}, data_args=dargs)


def validate(cfg: dict, hf_root=None, max_batches=None, random_init=True, allow_16bit=False) -> dict:
    import torch
    r = resolve(cfg, allow_16bit=allow_16bit)
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        raise RuntimeError("no GPU visible: the restoration path runs on MI355X only (no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    torch.manual_seed(r["seed"])
    from . import runner
    from .data import SyntheticImages
    from .dist import all_gather_images, broadcast_weights_sharded, shard_range
    lit = runner.LitUniFIE(r["model_kwargs"], dtype=r["dtype"], hf_root=hf_root, **r["caller_args"])
    no_ckpt = not any((r["model_kwargs"].get(k) or {}).get("ckpt_path") for k in ("frenc", "cnet", "tedit")) and not hf_root
    if no_ckpt and random_init:                           # no checkpoint reachable: seeded random weights of the architecture
        from .init import init_random_
        if rank == 0:
            init_random_(lit.model, r["seed"], "cpu")
    if world > 1:
        broadcast_weights_sharded(lit.model.to(dev), src=0)
    lit.model.refresh()
    data = SyntheticImages(**r["data_args"])
    if data.batch_size < world:
        raise ValueError(f"data batch_size {data.batch_size} < world size {world}: every rank needs at least one image per batch")
    sizes = [shard_range(data.batch_size, q, world)[1] - shard_range(data.batch_size, q, world)[0] for q in range(world)]
    n_img, secs, finite = 0, 0.0, True
    for i, batch in enumerate(data.batches(rank, world, dev)):
        if max_batches is not None and i >= max_batches:
            break
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        preds = lit.validation_step(batch, metrics=False)          # forward only inside the timed region
        out = preds[-1]
        if world > 1:
            out = all_gather_images(out, sizes)
        torch.cuda.synchronize()
        if i >= 1:                                        # batch 0 captures the hipGraph: time from the second one
            secs += time.perf_counter() - t0
            n_img += out.shape[0]
        finite = finite and bool(torch.isfinite(out).all())
        lit.update_metrics(preds[-1], batch[1])           # this rank's shard; reduced over the ranks below (fp64 CPU PSNR / SSIM: untimed)
    if world > 1:                                         # the reference's metric states reduce with dist_reduce_fx="sum"
        tot = torch.tensor([lit.totals["psnr"], lit.totals["ssim"], float(lit.totals["images"])], dtype=torch.float64, device=dev)
        dist.all_reduce(tot)
        lit.totals.update(psnr=float(tot[0]), ssim=float(tot[1]), images=int(tot[2]))
    res = dict(config=r["data_args"], dtype=r["dtype"], n_gpus=world, denoise_steps=r["model_kwargs"]["cnet"]["num_inference_steps"],
               images_per_s=(n_img / secs) if secs > 0 else None, output_finite=finite, **lit.metrics())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return res if rank == 0 else None


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m unirestore_amd.cli")
    ap.add_argument("command", choices=["validate", "print_config"])
    ap.add_argument("--config", required=True)
    ap.add_argument("--set", action="append", default=[], metavar="a.b.c=value", help="override a config key")
    ap.add_argument("--hf-root", default=None, help="folder with unet/ and vae/ diffusion_pytorch_model.safetensors (sd-turbo)")
    ap.add_argument("--max-batches", type=int, default=None)
    ap.add_argument("--allow-16bit", action="store_true", help="run a `precision: 32` config in fp16 (fp32 accumulation) instead of refusing it")
    a = ap.parse_args(argv)
    cfg = load_config(a.config, a.set)
    if a.command == "print_config":
        print(yaml.safe_dump(cfg, sort_keys=False))
        print(json.dumps(resolve(cfg, allow_16bit=a.allow_16bit)))
        return 0
    res = validate(cfg, hf_root=a.hf_root, max_batches=a.max_batches, allow_16bit=a.allow_16bit)
    if res is not None:
        print(json.dumps(res))
    return 0


if __name__ == "__main__":
    sys.exit(main())
