"""Caller side of the hot path: what `LitUniFIE.forward` and `ImageRestorationEvaluator.validation_step` do around
`DiffUIE.forward` (reference src/core/engine_unifie.py:227-236, src/core/base/eval_image_restoration.py:53-72,113-136),
without Lightning.  The centre crop is index arithmetic on the caller's tensor; resize / pad / un-pad / 8-bit
quantisation run as HIP kernels inside the model's graph (`DiffUIE.forward(..., quantize=True)`).
"""
from typing import List, Sequence

import torch


def crop_box(h: int, w: int, upper_h: int = 512, upper_w: int = 512):
    """crop_tensor's window (eval_image_restoration.py:113-136): (top, bottom, left, right) of the centre crop.
    Note the reference's `h//2 - crop_h//2 : h//2 + crop_h//2` drops one row/column when the crop size is odd."""
    ch, cw = min(h, upper_h), min(w, upper_w)
    return h // 2 - ch // 2, h // 2 + ch // 2, w // 2 - cw // 2, w // 2 + cw // 2


def crop_tensor(image: torch.Tensor) -> torch.Tensor:
    if image.ndim not in (3, 4):
        raise NotImplementedError                       # the reference raises here as well (:136)
    t, b, l, r = crop_box(image.shape[-2], image.shape[-1])
    return image[..., t:b, l:r]


@torch.inference_mode()
def forward(model, inputs: Sequence[torch.Tensor], task: str, quantize: bool = False) -> List[torch.Tensor]:
    """LitUniFIE.forward: inputs = [hq, lq] (any subset) -> [enh_hq, enh_lq]; one model.forward per list item."""
    return [model.forward(imgs, task, quantize=quantize) for imgs in inputs]


@torch.inference_mode()
def validation_step(model, lq: torch.Tensor, hq: torch.Tensor = None, task: str = "ir", need_crop: bool = True,
                    eval_types: Sequence[str] = ("lq",)):
    """Steps 1-2 of ImageRestorationEvaluator.validation_step: crop, restore, quantise to 8 bit.  Returns (preds, hq)."""
    if need_crop:
        lq = crop_tensor(lq) if "lq" in eval_types else lq
        hq = crop_tensor(hq) if (hq is not None and "hq" in eval_types) else hq
    inputs = ([hq] if "hq" in eval_types and hq is not None else []) + ([lq] if "lq" in eval_types else [])
    return forward(model, inputs, task, quantize=True), hq


def psnr(pred: torch.Tensor, target: torch.Tensor, data_range: float = 1.0) -> float:
    """Full-reference PSNR over the batch (what `val_lq/psnr` reports, eval_image_restoration.py:102-104)."""
    mse = torch.mean((pred.double().cpu() - target.double().cpu()) ** 2)
    return float(10.0 * torch.log10(data_range ** 2 / mse))
