"""Synthetic inputs for the config entry points (the reference's datasets / corruption pipeline are out of scope).

`SyntheticImages` yields the evaluator's batch tuple `(lq, hq, gt, fname, task)` (reference
src/core/base/eval_image_restoration.py:56) from seeded random images: hq = torch.rand (what the reference's own smoke
test feeds, src/modules/diffuie/autoencoder.py:209), lq = hq under one of three cost-neutral degradations
(SURVEY.md 8d): additive N(0, 0.1) noise, haze 0.5*x + 0.5, low light 0.3*x.  Content does not change the cost of the path.
"""
from typing import Iterator, List, Sequence, Tuple

import torch

DEGRADATIONS = ("noise", "haze", "lowlight")


def degrade(hq: torch.Tensor, kind: str, generator: torch.Generator = None) -> torch.Tensor:
    if kind == "noise":
        return (hq + 0.1 * torch.randn(hq.shape, generator=generator, device=hq.device)).clamp(0, 1)
    if kind == "haze":
        return 0.5 * hq + 0.5
    if kind == "lowlight":
        return 0.3 * hq
    raise ValueError(f"unknown degradation {kind!r}: choose from {DEGRADATIONS}")


class SyntheticImages:
    def __init__(self, task: str = "ir", resolution: Sequence[int] = (512, 512), batch_size: int = 8, num_batches: int = 5,
                 degradations: Sequence[str] = ("noise",), seed: int = 42):
        for d in degradations:
            if d not in DEGRADATIONS:
                raise ValueError(f"unknown degradation {d!r}: choose from {DEGRADATIONS}")
        if isinstance(resolution, int):
            resolution = (resolution, resolution)
        self.task, self.resolution, self.batch_size, self.num_batches = task, tuple(resolution), int(batch_size), int(num_batches)
        self.degradations, self.seed = list(degradations), seed

    def __len__(self):
        return self.num_batches

    def batches(self, rank: int = 0, world: int = 1, device="cpu") -> Iterator[Tuple[torch.Tensor, torch.Tensor, None, List[str], str]]:
        """Rank `rank`'s contiguous shard of every global batch (sizes differ by at most one image)."""
        from .dist import shard_range
        h, w = self.resolution
        lo, hi = shard_range(self.batch_size, rank, world)
        for b in range(self.num_batches):
            g = torch.Generator().manual_seed(self.seed + b)          # the GLOBAL batch is the same whatever the world size
            hq = torch.rand(self.batch_size, 3, h, w, generator=g)
            lq = torch.stack([degrade(hq[i], self.degradations[i % len(self.degradations)], g) for i in range(self.batch_size)])
            names = [f"syn_{b:04d}_{i:03d}" for i in range(lo, hi)]
            yield lq[lo:hi].to(device), hq[lo:hi].to(device), None, names, self.task
