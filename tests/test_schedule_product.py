"""Row S of SURVEY.md 8(a) on the PRODUCT's schedule module (unirestore_amd.schedule, what the HIP path reads its
timesteps / coefficients from): integer DDIM schedule and fp32 alpha-bar table bit-exact against the closed-form known
answers of diffusers' DDIMScheduler(sd-turbo config) - /root/reference/src/modules/diffuie/unifie.py:65-75,150 - and the
fused DDIM coefficients against the oracle's step.  CPU only (the module is pure host code)."""
import numpy as np
import torch

from unirestore_amd import schedule as ps

KAT_ALPHA = {0: 0.9991499781608582, 49: 0.9526252746582031, 249: 0.6754320859909058, 499: 0.27766942977905273,
             749: 0.05662344768643379, 999: 0.00466009508818388}


def test_integer_schedule_bit_exact():
    assert ps.ddim_timesteps(1).tolist() == [999]
    assert ps.ddim_timesteps(4).tolist() == [999, 749, 499, 249]                      # == train_timesteps, unifie.py:67
    assert ps.ddim_timesteps(20).tolist() == list(range(999, 0, -50))
    assert ps.ddim_timesteps(50).tolist() == list(range(999, 0, -20))
    for n in (1, 2, 4, 20, 50):
        t = ps.ddim_timesteps(n)
        assert t.dtype == np.int64 and len(t) == n and t[0] == 999 and (np.diff(t) < 0).all()


def test_alpha_bar_table_bit_exact():
    ac = ps.alphas_cumprod()
    assert ac.dtype == torch.float32 and ac.shape == (1000,)
    for t, v in KAT_ALPHA.items():
        assert float(ac[t]) == v, (t, float(ac[t]), v)                                # exact fp32 equality
    from oracle import schedule as osched
    assert torch.equal(ac, osched.alphas_cumprod())
    for n in (1, 4, 20, 50):
        assert np.array_equal(ps.ddim_timesteps(n), osched.ddim_timesteps(n))


def test_ddim_coefficients_match_the_scheduler_step():
    """zt' = c_x*zt + c_e*eps must equal DDIMScheduler.step (eta 0, no clipping, set_alpha_to_one=False)."""
    from oracle import schedule as osched
    g = torch.Generator().manual_seed(0)
    x, e = torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64), torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64)
    for n in (1, 4, 20, 50):
        for t in ps.ddim_timesteps(n):
            c_x, c_e = ps.ddim_coefficients(int(t), n)
            ref = osched.ddim_step(e, int(t), x, n)
            got = c_x * x + c_e * e
            assert float((got - ref).abs().max()) < 2e-6 * float(ref.abs().max()), (n, int(t))   # fp32 table, fp64 algebra
    # last step of every schedule lands on alpha_bar[0], not 1.0
    c_x, c_e = ps.ddim_coefficients(249, 4)
    ac = ps.alphas_cumprod_f64()
    assert abs(c_x - (ac[0] / ac[249]) ** 0.5) < 1e-12
