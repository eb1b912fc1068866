"""The thin C-ABI entry points of include/unirestore_hip.h that the module graph reaches through richer paths (SURVEY.md 8b names
them): each one is LAUNCHED here through ctypes and compared with plain PyTorch fp32 on 16-bit-rounded inputs, in both types.
Tolerances: 16-bit outputs 3e-3 (bf16) / 4e-4 (fp16) rel-L2; fp32 outputs 2e-4 (as tests/test_ops_gpu.py)."""
import math

import pytest
import torch
import torch.nn.functional as F

from golden_util import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["bf16", "fp16"])
def env(request):
    from unirestore_amd import capi, ops
    dt = ops.set_dtype(request.param)
    code = capi.UR_DT_BF16 if request.param == "bf16" else capi.UR_DT_F16
    yield capi, dt, code, (3e-3 if request.param == "bf16" else 4e-4)
    ops.set_dtype("bf16")


def _s():
    return torch.cuda.current_stream().cuda_stream


def test_gemm_bias_act(env):
    """ur_gemm_bias_act: y = act(x W^T + b) (+ residual) - Linear / 1x1 conv with the SURVEY 8(b) epilogues."""
    capi, dt, code, tol = env
    g = torch.Generator().manual_seed(1)
    m, n, k = 384, 192, 256
    x = torch.randn(m, k, generator=g).to(dt)
    w = (torch.randn(n, k, generator=g) / math.sqrt(k)).to(dt)
    b = torch.randn(n, generator=g)
    r = torch.randn(m, n, generator=g).to(dt)
    ws = torch.empty(1 << 22, dtype=torch.float32, device="cuda")
    for act, fn in ((capi.UR_ACT_NONE, lambda t: t), (capi.UR_ACT_SILU, F.silu), (capi.UR_ACT_GELU, F.gelu)):
        y = torch.empty(m, n, dtype=dt, device="cuda")
        xd, wd, bd, rd = x.cuda(), w.cuda(), b.cuda(), r.cuda()
        capi.check(capi.lib.ur_gemm_bias_act(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), rd.data_ptr(), y.data_ptr(), m, n, k, k, k, n, n, act,
                                             ws.data_ptr(), ws.numel() * 4, code, _s()))
        ref = fn(x.float() @ w.float().t() + b) + r.float()
        assert rel_l2(y.float().cpu(), ref) < tol, act


def test_groupconv3x3(env):
    """ur_groupconv3x3_nhwc: AdaNAFV2.group_conv / the fused TFA gate branches (cfrm.py:20-21, taskeditor.py:30-52)."""
    capi, dt, code, tol = env
    g = torch.Generator().manual_seed(2)
    n, h, w_, groups, cg, cog = 2, 16, 24, 4, 32, 48
    x = torch.randn(n, groups * cg, h, w_, generator=g).to(dt)
    wt = (torch.randn(groups * cog, cg, 3, 3, generator=g) / math.sqrt(9 * cg)).to(dt)
    b = torch.randn(groups * cog, generator=g)
    ref = F.gelu(F.conv2d(x.float(), wt.float(), b, padding=1, groups=groups))
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    wd = wt.permute(0, 2, 3, 1).reshape(groups * cog, 9 * cg).contiguous().cuda()         # [G*Cog][kh, kw, Cg]
    y = torch.empty(n, h, w_, groups * cog, dtype=dt, device="cuda")
    ws = torch.empty(1 << 22, dtype=torch.float32, device="cuda")
    bd = b.cuda()                                                                              # (kept alive across the launch)
    capi.check(capi.lib.ur_groupconv3x3_nhwc(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), n, h, w_, cg, cog, groups,
                                             capi.UR_ACT_GELU, ws.data_ptr(), ws.numel() * 4, code, _s()))
    assert rel_l2(y.float().cpu().permute(0, 3, 1, 2), ref) < tol


def test_groupnorm_nhwc_instnorm_avgpool_and_size_queries(env):
    """ur_groupnorm_nhwc (statistics -> finalize -> apply chained, two virtually concatenated sources), ur_instnorm_stats,
    ur_avgpool_hw, ur_groupnorm_ws_bytes / ur_groupnorm_ab_bytes."""
    capi, dt, code, tol = env
    lib = capi.lib
    g = torch.Generator().manual_seed(3)
    n, hw, c1, c2, groups = 3, 40 * 24, 64, 32, 8
    x1 = (torch.randn(n, hw, c1, generator=g) * 2 + 0.5).to(dt)
    x2 = (torch.randn(n, hw, c2, generator=g) - 1.0).to(dt)
    gamma, beta = 1 + 0.3 * torch.randn(c1 + c2, generator=g), 0.2 * torch.randn(c1 + c2, generator=g)
    p1, p2 = lib.ur_groupnorm_stats_parts(n, hw, c1), lib.ur_groupnorm_stats_parts(n, hw, c2)
    assert p1 > 0 and lib.ur_groupnorm_ws_bytes(n, hw, c1) == n * p1 * c1 * 2 * 4 and lib.ur_groupnorm_ab_bytes(n, c1 + c2) == n * (c1 + c2) * 2 * 4
    ws = torch.empty((lib.ur_groupnorm_ws_bytes(n, hw, c1) + lib.ur_groupnorm_ws_bytes(n, hw, c2)) // 4, dtype=torch.float32, device="cuda")
    ab = torch.empty(lib.ur_groupnorm_ab_bytes(n, c1 + c2) // 4, dtype=torch.float32, device="cuda")
    y = torch.empty(n, hw, c1 + c2, dtype=dt, device="cuda")
    x1d, x2d, gd, bd = x1.cuda(), x2.cuda(), gamma.cuda(), beta.cuda()                        # (kept alive across the launch)
    capi.check(lib.ur_groupnorm_nhwc(x1d.data_ptr(), x2d.data_ptr(), y.data_ptr(), gd.data_ptr(), bd.data_ptr(), n, hw, c1, c2,
                                     groups, 1e-5, 1, ws.data_ptr(), ab.data_ptr(), None, 0, None, 0, code, _s()))
    cat = torch.cat([x1.float(), x2.float()], -1).permute(0, 2, 1)                      # [N, C, HW]
    ref = F.silu(F.group_norm(cat, groups, gamma, beta, 1e-5)).permute(0, 2, 1)
    assert rel_l2(y.float().cpu(), ref) < tol
    # InstanceNorm statistics (= per-channel sums) and the global average pool on the same plane layout
    part = torch.empty(n, p1, c1, 2, dtype=torch.float32, device="cuda")
    capi.check(lib.ur_instnorm_stats(x1d.data_ptr(), part.data_ptr(), n, hw, c1, code, _s()))
    s = part.sum(1).cpu()
    assert torch.allclose(s[..., 0], x1.float().sum(1), rtol=1e-4, atol=1e-2)
    assert torch.allclose(s[..., 1], (x1.float() ** 2).sum(1), rtol=1e-4, atol=1e-2)
    pooled = torch.empty(n, c1, dtype=torch.float32, device="cuda")
    wsp = torch.empty(lib.ur_groupnorm_ws_bytes(n, hw, c1) // 4, dtype=torch.float32, device="cuda")
    capi.check(lib.ur_avgpool_hw(x1d.data_ptr(), pooled.data_ptr(), n, hw, c1, wsp.data_ptr(), code, _s()))
    assert rel_l2(pooled.cpu(), x1.float().mean(1)) < 2e-4
