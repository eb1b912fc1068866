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


def psnr_per_image(pred: torch.Tensor, target: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    """PSNR of every image of the batch (fp64 [N]): skimage's peak_signal_noise_ratio per sample - what the reference's SKPSNR sums
    before dividing by the image count (eval_image_restoration.py:266-278)."""
    d = (pred.double().cpu() - target.double().cpu()) ** 2
    mse = d.reshape(d.shape[0], -1).mean(dim=1)
    return 10.0 * torch.log10(data_range ** 2 / mse)


def psnr(pred: torch.Tensor, target: torch.Tensor, data_range: float = 1.0) -> float:
    """`val_lq/psnr` of a batch: the MEAN OF THE PER-IMAGE PSNRs (not the PSNR of the batch-mean MSE: the two differ as soon as
    the images differ; eval_image_restoration.py:102-104,181,266-278)."""
    return float(psnr_per_image(pred, target, data_range).mean())


def ssim(pred: torch.Tensor, target: torch.Tensor, data_range: float = 1.0, win: int = 7) -> float:
    """Mean structural similarity, scikit-image's defaults (`structural_similarity`: uniform 7x7 window, K1 0.01, K2 0.03,
    sample covariance, borders of (win-1)//2 pixels dropped), averaged over channels and the batch - what the reference's
    `SKSSIM(data_range=1.0)` metric reports (eval_image_restoration.py:182)."""
    import torch.nn.functional as F
    x, y = pred.double().cpu(), target.double().cpu()
    c = x.shape[1]
    k = torch.full((c, 1, win, win), 1.0 / (win * win), dtype=torch.float64)
    filt = lambda t: F.conv2d(t, k, groups=c)                     # 'valid' = the cropped interior skimage averages over
    npx = win * win
    cov_norm = npx / (npx - 1.0)
    ux, uy = filt(x), filt(y)
    vx = cov_norm * (filt(x * x) - ux * ux)
    vy = cov_norm * (filt(y * y) - uy * uy)
    vxy = cov_norm * (filt(x * y) - ux * uy)
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
    return float(s.mean())


class LitUniFIE:
    """The caller the reference builds around the path (`LitUniFIE{IR,MTL}`, src/core/engine_unifie.py:29-42,227-236 +
    ImageRestorationEvaluator.validation_step) without Lightning: owns a `DiffUIE`, maps `forward` over the input list and
    runs the evaluator's crop / restore / quantise / metric steps on a batch tuple `(lq, hq, gt, fname, task)`."""

    def __init__(self, model_kwargs: dict, save_image: bool = False, eval_mode: str = "FR", need_crop: bool = True,
                 dtype: str = "bf16", hf_root: str = None, model=None, **_ignored):
        from . import checkpoint
        self.model_kwargs, self.need_crop, self.eval_mode = model_kwargs, need_crop, eval_mode
        tedit = model_kwargs.get("tedit") or {}
        self.task_dict = tedit.get("task", [])
        self.model = model if model is not None else checkpoint.build_from_config(model_kwargs, hf_root=hf_root, dtype=dtype)
        self.totals = dict(psnr=0.0, ssim=0.0, images=0)

    def forward(self, inputs: Sequence[torch.Tensor], task: str, quantize: bool = False) -> List[torch.Tensor]:
        return forward(self.model, inputs, task, quantize=quantize)

    def validation_step(self, batch, eval_types: Sequence[str] = ("lq",), metrics: bool = True):
        lq, hq, _gt, _fname, _task = batch
        # the IR evaluator always restores with the "ir" prompt (eval_image_restoration.py:70: self.forward(inputs, 'ir')), whatever
        # task tag the batch carries; task-driven decoding belongs to the downstream (MTL) evaluators, which are out of scope
        preds, hq_c = validation_step(self.model, lq, hq, task="ir", need_crop=self.need_crop, eval_types=eval_types)
        if metrics and hq_c is not None and self.eval_mode in ("FR", "ALL") and preds[-1].shape == (crop_tensor(hq) if self.need_crop else hq).shape:
            tgt = crop_tensor(hq) if self.need_crop else hq
            n = preds[-1].shape[0]
            self.totals["psnr"] += float(psnr_per_image(preds[-1], tgt).sum())          # sum over images (SKPSNR state)
            self.totals["ssim"] += ssim(preds[-1], tgt) * n
            self.totals["images"] += n
        return preds

    # ---- the training step's FORWARD halves (engine_unifie.py:135-191), values only: the HIP path has no autograd ------------
    @torch.no_grad()
    def fr_training_fwd(self, hq, lq):
        """AE encoder (+CFRM on the degraded input): (h0, h0_mids, l0, l0_mids) - engine_unifie.py:135-148."""
        h0, h0_mids = self.model.ae.encode(hq, enable_fr=False, noise=self._noise(hq))        # (ae.encode re-asserts the model's dtype)
        l0, l0_mids = self.model.ae.encode(lq, enable_fr=bool(self.model_kwargs.get("frenc")), noise=self._noise(lq))
        return h0, h0_mids, l0, l0_mids

    @staticmethod
    def fr_loss_fn(h0, h0_mids, l0, l0_mids):
        """Feature-MSE taps of the CFRM stage: 0.1 * L1 + 0.1 * L2 + 0.01 * L3 (+ the latent MSE it logs) - :150-168."""
        mse = lambda a, b: torch.mean((a.float() - b.float()) ** 2)
        layers = [mse(a, b) for a, b in zip(l0_mids, h0_mids)]
        return dict(loss_layer1=layers[0], loss_layer2=layers[1], loss_layer3=layers[2], loss_enc=mse(l0, h0),
                    loss_frenc=0.1 * layers[0] + 0.1 * layers[1] + 0.01 * layers[2])

    @torch.no_grad()
    def cn_training_fwd(self, h0, l0, timesteps=None, noise=None):
        """Controller + SC-Tuner + UNet: diffuse the clean latent, predict z0 under the degraded latent's control - :170-178."""
        zt, _, ts = self.model.diffuse(h0, timesteps, noise)
        return self.model.predict_z0(zt, conditions=l0, timesteps=ts)

    @staticmethod
    def cn_loss_fn(pred_z0, h0):
        return torch.mean((pred_z0.float() - h0.float().to(pred_z0.device)) ** 2)

    @torch.no_grad()
    def te_training_fwd(self, pred_z0, l0_mids, task):
        """AE decoder + TFA on the predicted latent - :186-191."""
        return self.model.ae.decode(pred_z0, l0_mids, task)

    noise_seed = None          # set to an int to make the VAE's posterior sampling reproducible (tests)

    def _noise(self, img):
        if self.noise_seed is None:
            return None
        g = torch.Generator().manual_seed(self.noise_seed + int(img.shape[-1]))
        return torch.randn(img.shape[0], self.model.ae.vae.latent_channels, img.shape[-2] // 8, img.shape[-1] // 8, generator=g)

    def update_metrics(self, preds: torch.Tensor, hq: torch.Tensor):
        """Metric states for one batch of restored images vs their clean targets (for callers that time the forward separately)."""
        tgt = crop_tensor(hq) if self.need_crop else hq
        if self.eval_mode in ("FR", "ALL") and preds.shape == tgt.shape:
            n = preds.shape[0]
            self.totals["psnr"] += float(psnr_per_image(preds, tgt).sum())
            self.totals["ssim"] += ssim(preds, tgt) * n
            self.totals["images"] += n

    def metrics(self) -> dict:
        n = max(self.totals["images"], 1)
        return {"val_lq/psnr": self.totals["psnr"] / n, "val_lq/ssim": self.totals["ssim"] / n, "images": self.totals["images"]}
