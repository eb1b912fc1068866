"""Timestep schedule + DDPM/DDIM arithmetic (oracle; test infrastructure only).

Follows the scheduler calls at /root/reference/src/modules/diffuie/unifie.py:69-75
(from_pretrained + set_timesteps), :88 (add_noise), :150 (DDIM step) with the
sd-turbo scheduler config (scaled_linear betas 0.00085..0.012, 1000 train steps,
"trailing" spacing, epsilon prediction, eta=0, no clipping, set_alpha_to_one=False).
"""
import numpy as np
import torch

NUM_TRAIN_TIMESTEPS = 1000
BETA_START, BETA_END = 0.00085, 0.012


def alphas_cumprod() -> torch.Tensor:
    betas = torch.linspace(BETA_START ** 0.5, BETA_END ** 0.5, NUM_TRAIN_TIMESTEPS, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def ddim_timesteps(num_inference_steps: int) -> np.ndarray:
    """'trailing' spacing: round(arange(T, 0, -T/N)) - 1 (int64)."""
    step_ratio = NUM_TRAIN_TIMESTEPS / num_inference_steps
    return np.round(np.arange(NUM_TRAIN_TIMESTEPS, 0, -step_ratio)).astype(np.int64) - 1


def add_noise(x0: torch.Tensor, noise: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    ac = alphas_cumprod().to(x0.device)
    a = ac[t].to(x0.dtype)
    sa = (a ** 0.5).view(-1, *([1] * (x0.dim() - 1)))
    sb = ((1 - a) ** 0.5).view(-1, *([1] * (x0.dim() - 1)))
    return sa * x0 + sb * noise


def ddim_step(eps: torch.Tensor, t: int, x: torch.Tensor, num_inference_steps: int) -> torch.Tensor:
    ac = alphas_cumprod().to(x.device)
    t_prev = int(t) - NUM_TRAIN_TIMESTEPS // num_inference_steps
    a_t = ac[int(t)]
    a_p = ac[t_prev] if t_prev >= 0 else ac[0]      # set_alpha_to_one=False
    x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
    return a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * eps
