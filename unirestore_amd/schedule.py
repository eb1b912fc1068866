"""DDIM / DDPM schedule tables for the hot path (host side; integer schedule bit-exact with the reference).

Follows the scheduler calls at /root/reference/src/modules/diffuie/unifie.py:69-75,88,150 with the sd-turbo
scheduler config (scaled_linear betas 0.00085..0.012 over 1000 steps, "trailing" spacing, epsilon prediction,
eta = 0, no clipping, set_alpha_to_one = False) — SURVEY.md Appendix C.7.
"""
import numpy as np
import torch

NUM_TRAIN_TIMESTEPS = 1000
BETA_START, BETA_END = 0.00085, 0.012


def alphas_cumprod() -> torch.Tensor:
    """fp32 table, computed exactly as diffusers does (fp32 linspace of sqrt(beta), squared, fp32 cumprod)."""
    betas = torch.linspace(BETA_START ** 0.5, BETA_END ** 0.5, NUM_TRAIN_TIMESTEPS, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def alphas_cumprod_f64() -> np.ndarray:
    return alphas_cumprod().numpy().astype(np.float64)


def ddim_timesteps(num_inference_steps: int) -> np.ndarray:
    step_ratio = NUM_TRAIN_TIMESTEPS / num_inference_steps
    return np.round(np.arange(NUM_TRAIN_TIMESTEPS, 0, -step_ratio)).astype(np.int64) - 1


def ddim_coefficients(t: int, num_inference_steps: int):
    """x_prev = c_x * x_t + c_e * eps  (x0 = (x_t - sqrt(1-a_t) eps)/sqrt(a_t); x_prev = sqrt(a_p) x0 + sqrt(1-a_p) eps)."""
    ac = alphas_cumprod_f64()
    t_prev = int(t) - NUM_TRAIN_TIMESTEPS // num_inference_steps
    a_t = ac[int(t)]
    a_p = ac[t_prev] if t_prev >= 0 else ac[0]
    c_x = (a_p / a_t) ** 0.5
    c_e = (1.0 - a_p) ** 0.5 - c_x * (1.0 - a_t) ** 0.5
    return float(c_x), float(c_e)
