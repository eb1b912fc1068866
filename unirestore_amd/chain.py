"""Host side of the token-stationary fused chains (csrc/tchain.hip): weight-stream packers + op wrappers.

A chain kernel streams its weights as 42-KiB tiles that are byte-for-byte the LDS image the MFMA fragment reads expect
(csrc/tchain.hip header): [rows][128 B] blocks of 64 K-values per row with the 16-byte slot XOR-swizzled by (row >> 1) & 7,
output rows permuted by swap23 inside every 32-row block (so that a stage's accumulator registers ARE the next stage's B
fragments), LayerNorm folded into the consuming weights, fp32 epilogue vectors in the 2-KiB tail of the tile.
The packers run once per (module, 16-bit type); results are cached by the modules like every other packed weight.
"""
import torch

from . import ops
from .capi import check, lib

TILE_W = 40960
TILE_AUX = 4096
TILE = TILE_W + TILE_AUX
CHAIN_C = 320            # channels the kernels are instantiated for (UNet level 0)
CHAIN_TOK = 128          # tokens per workgroup


def _swap23(i: int) -> int:
    return (i & 0x13) | ((i & 4) << 1) | ((i & 8) >> 1)


_PERM32 = torch.tensor([_swap23(i) for i in range(32)])


def perm_rows(n: int) -> torch.Tensor:
    """image row -> source row for an n-row matrix (n % 32 == 0): MFMA row i of every 32-row block holds channel swap23(i)."""
    assert n % 32 == 0
    return (torch.arange(n) // 32) * 32 + _PERM32.repeat(n // 32)


def lds_block(w16: torch.Tensor) -> torch.Tensor:
    """[R, 64] 16-bit -> uint8 [R * 128]: row-major 128-byte rows, logical 16-byte slot s stored at slot s ^ ((row >> 1) & 7)."""
    r = w16.shape[0]
    assert w16.shape[1] == 64 and w16.element_size() == 2
    v = w16.reshape(r, 8, 8)
    rows = torch.arange(r, device=w16.device)
    phys = torch.arange(8, device=w16.device)[None, :] ^ ((rows >> 1) & 7)[:, None]
    return v[rows[:, None], phys].contiguous().reshape(-1).view(torch.uint8)


def _tile(blocks, aux: torch.Tensor = None) -> torch.Tensor:
    """weight blocks (uint8 tensors, concatenated, <= 40 KiB) + fp32 aux vector (<= 512 floats) -> one TILE-byte tile"""
    t = torch.zeros(TILE, dtype=torch.uint8, device=blocks[0].device)
    w = torch.cat(blocks)
    assert w.numel() <= TILE_W
    t[:w.numel()] = w
    if aux is not None:
        a = aux.to(torch.float32).contiguous().view(torch.uint8)
        assert a.numel() <= TILE_AUX
        t[TILE_W:TILE_W + a.numel()] = a
    return t


def gemm_tiles(w: torch.Tensor, dt, aux_last: torch.Tensor = None, aux_at=None):
    """Stage y = W x for W [N, K] fp32 (N <= 320, N % 32 == 0, K % 64 == 0): K/64 tiles of [N rows][128 B], rows swap23-permuted.
    aux_last: fp32 vector placed in the aux area of the last tile; aux_at: {tile index: vector} for others."""
    n, k = w.shape
    assert n % 32 == 0 and k % 64 == 0 and n * 128 <= TILE_W
    w16 = w[perm_rows(n).to(w.device)].to(dt)
    tiles = []
    for kt in range(k // 64):
        aux = aux_last if kt == k // 64 - 1 else None
        if aux_at and kt in aux_at:
            aux = aux_at[kt]
        tiles.append(_tile([lds_block(w16[:, 64 * kt:64 * kt + 64])], aux))
    return tiles


def fold_ln(w: torch.Tensor, b, gamma: torch.Tensor, beta: torch.Tensor, dt):
    """Linear(LayerNorm(x)) -> (w' = W * gamma, bias' = W . beta + b, colsum[n] = sum_k dt(w'[n, k])) (all fp32; w' not yet rounded)"""
    wf = w * gamma[None, :]
    bf = w @ beta + (b if b is not None else 0.0)
    return wf, bf, wf.to(dt).float().sum(dim=1)


def pack_mlp(w1, b1, w2, b2, gamma, beta, dev, dt=None) -> torch.Tensor:
    """FeedForward(GEGLU) with its LayerNorm folded: w1 [2H, C] (rows [0, H) = value, [H, 2H) = gate: diffusers GEGLU.chunk),
    w2 [C, H].  Stream per 64 hidden units: FF1 tile (units 0..31), FF1 tile (32..63), FF2 tile (K slice of 64)."""
    dt = dt or ops.act_dtype()
    w1, b1, w2, b2, gamma, beta = (t.detach().to(dev, torch.float32) for t in (w1, b1, w2, b2, gamma, beta))
    h2, c = w1.shape
    hid = h2 // 2
    assert c == CHAIN_C and w2.shape == (c, hid) and hid % 64 == 0
    wf, bf, cs = fold_ln(w1, b1, gamma, beta, dt)
    wf16 = wf.to(dt)
    w2p = w2[perm_rows(c).to(dev)].to(dt)
    p32 = _PERM32.to(dev)
    tiles = []
    for ch in range(hid // 64):
        for half in range(2):
            u0 = ch * 64 + half * 32
            ra, rg = u0 + p32, hid + u0 + p32                     # image rows: a block then g block, swap23 order
            blocks = [lds_block(torch.cat([wf16[ra, 64 * kb:64 * kb + 64], wf16[rg, 64 * kb:64 * kb + 64]], 0)) for kb in range(c // 64)]
            aux = torch.cat([bf[u0:u0 + 32], bf[hid + u0:hid + u0 + 32], cs[u0:u0 + 32], cs[hid + u0:hid + u0 + 32]])   # natural order
            tiles.append(_tile(blocks, aux))
        tiles.append(_tile([lds_block(w2p[:, 64 * ch:64 * ch + 64])], b2 if ch == hid // 64 - 1 else None))
    return torch.cat(tiles).contiguous()


def _ln_aux(bias: torch.Tensor, colsum: torch.Tensor) -> torch.Tensor:
    """aux area of a LayerNorm-folded stage: bias' at floats 0.., column sums at floats 512.."""
    a = torch.zeros(1024, dtype=torch.float32, device=bias.device)
    a[:bias.numel()] = bias
    a[512:512 + colsum.numel()] = colsum
    return a


def pack_head(w_in, b_in, wq, wk, wv, gamma, beta, dev, dt=None) -> torch.Tensor:
    """Transformer2DModel.proj_in + attn1.to_q / to_k / to_v over LayerNorm1 (folded): 4 stages of 5 tiles (C = 320)."""
    dt = dt or ops.act_dtype()
    w_in, b_in, wq, wk, wv, gamma, beta = (t.detach().to(dev, torch.float32) for t in (w_in, b_in, wq, wk, wv, gamma, beta))
    assert w_in.shape == (CHAIN_C, CHAIN_C)
    tiles = gemm_tiles(w_in, dt, aux_last=b_in)
    for w in (wq, wk, wv):
        wf, bf, cs = fold_ln(w, None, gamma, beta, dt)
        tiles += gemm_tiles(wf, dt, aux_last=_ln_aux(bf, cs))
    return torch.cat(tiles).contiguous()


def pack_tail(wo1, bo1, wq2, g2, be2, wk2, wv2, ctx, wo2, bo2, w1, b1, w2, b2, g3, be3, w_out, b_out, heads, dev, dt=None) -> torch.Tensor:
    """Everything behind the self-attention of a BasicTransformerBlock + proj_out (see csrc/tchain.hip, kind TAIL).
    ctx: the constant cross-attention context [Tk, cross_dim] (Tk <= 80); its K / V projections are baked into the stream."""
    dt = dt or ops.act_dtype()
    f = lambda t: t.detach().to(dev, torch.float32)
    wo1, bo1, wq2, g2, be2, wk2, wv2, ctx, wo2, bo2, g3, be3, w_out, b_out = map(f, (wo1, bo1, wq2, g2, be2, wk2, wv2, ctx, wo2, bo2, g3, be3, w_out, b_out))
    c = CHAIN_C
    d = c // heads
    tk = ctx.shape[0]
    assert heads * d == c and d == 64 and tk <= 80 and wq2.shape == (c, c)
    tiles = gemm_tiles(wo1, dt, aux_last=bo1)
    wf, bf, cs = fold_ln(wq2, None, g2, be2, dt)
    wf16 = wf.to(dt)
    # context keys / values: computed from the 16-bit-rounded operands (as the per-layer path does), stored in the 16-bit type
    ctx16 = ctx.to(dt).float()
    kc = (ctx16 @ wk2.to(dt).float().t()).to(dt)                    # [Tk, C]
    vc = (ctx16 @ wv2.to(dt).float().t()).to(dt)                    # [Tk, C]
    kpad = torch.zeros(96, c, dtype=dt, device=dev)
    kpad[:tk] = kc
    vtp = torch.zeros(c, 128, dtype=dt, device=dev)
    vtp[:, :tk] = vc.t()
    pk, pd, p32 = perm_rows(96).to(dev), perm_rows(d).to(dev), _PERM32.to(dev)
    out2 = gemm_tiles(wo2, dt, aux_last=bo2)                        # K slice hd of to_out2 = its tile hd
    for hd in range(heads):
        ra, rb = d * hd + p32, d * hd + 32 + p32
        qblocks = [lds_block(torch.cat([wf16[ra, 64 * kb:64 * kb + 64], wf16[rb, 64 * kb:64 * kb + 64]], 0)) for kb in range(c // 64)]
        tiles.append(_tile(qblocks, _ln_aux(bf[d * hd:d * hd + d], cs[d * hd:d * hd + d])))
        tiles.append(_tile([lds_block(kpad[pk][:, d * hd:d * hd + d].contiguous()),
                            lds_block(vtp[d * hd + pd][:, 0:64].contiguous()), lds_block(vtp[d * hd + pd][:, 64:128].contiguous())]))
        tiles.append(out2[hd])
    mlp = pack_mlp(w1, b1, w2, b2, g3, be3, dev, dt)
    tiles.append(mlp)
    tiles += gemm_tiles(w_out, dt, aux_last=b_out)
    return torch.cat(tiles).contiguous()


def pack_csce(w_proj, b_proj, w0, b0, w2, b2, dev, dt=None) -> torch.Tensor:
    """CSCEAdapter (scedit.py:24-38) for a 320-channel skip and a 256-channel condition: proj (K = 256: 4 tiles), tuner.0, tuner.2."""
    dt = dt or ops.act_dtype()
    f = lambda t: t.detach().to(dev, torch.float32).reshape(t.shape[0], -1) if t.dim() > 1 else t.detach().to(dev, torch.float32)
    w_proj, b_proj, w0, b0, w2, b2 = map(f, (w_proj, b_proj, w0, b0, w2, b2))
    assert w_proj.shape == (CHAIN_C, 256) and w0.shape == (CHAIN_C, CHAIN_C) and w2.shape == (CHAIN_C, CHAIN_C)
    tiles = gemm_tiles(w_proj, dt, aux_last=b_proj) + gemm_tiles(w0, dt, aux_last=b0) + gemm_tiles(w2, dt, aux_last=b2)
    return torch.cat(tiles).contiguous()


def csce_fused(x: torch.Tensor, cond: torch.Tensor, stream_w: torch.Tensor, gn=True) -> torch.Tensor:
    """x [N,H,W,320], cond [N,H,W,256] 16-bit -> edited skip [N,H,W,320] (+ ._gn partial plane)"""
    n, c = x.shape[0], x.shape[-1]
    rows = x.numel() // c
    hw = rows // n
    y = torch.empty_like(x)
    part = torch.empty((n, hw // CHAIN_TOK, c, 2), dtype=torch.float32, device=x.device) if gn else None
    check(lib.ur_csce_fused(x.data_ptr(), cond.data_ptr(), stream_w.data_ptr(), stream_w.numel(), y.data_ptr(),
                            None if part is None else part.data_ptr(), rows, hw, c, cond.shape[-1], ops._dt(x), ops._stream()))
    if gn:
        y._gn = (part, hw // CHAIN_TOK)
    return y


def chain_ok(x: torch.Tensor) -> bool:
    """The chain kernels cover C = 320 with whole 128-token tiles."""
    rows = x.numel() // x.shape[-1]
    return x.shape[-1] == CHAIN_C and rows % CHAIN_TOK == 0 and x.is_contiguous()


def transformer_head_fused(x: torch.Tensor, ab: torch.Tensor, stream_w: torch.Tensor, n_img: int, eps: float):
    """x [N*HW, 320] (or [N,H,W,320]) 16-bit, ab fp32 [N][2][320] -> (h0 [T,320], q [N,HW,320], k [N,HW,320], v^T [N,320,HW])"""
    c = x.shape[-1]
    rows = x.numel() // c
    hw = rows // n_img
    h0 = torch.empty((rows, c), dtype=x.dtype, device=x.device)
    q = torch.empty((n_img, hw, c), dtype=x.dtype, device=x.device)
    k = torch.empty((n_img, hw, c), dtype=x.dtype, device=x.device)
    vt = torch.empty((n_img, c, hw), dtype=x.dtype, device=x.device)
    check(lib.ur_transformer_head_fused(x.data_ptr(), ab.data_ptr(), stream_w.data_ptr(), stream_w.numel(), h0.data_ptr(), q.data_ptr(),
                                        k.data_ptr(), vt.data_ptr(), rows, hw, c, eps, ops._dt(x), ops._stream()))
    return h0, q, k, vt


def transformer_tail_fused(o1, h0, xres, stream_w, n_img: int, hidden: int, heads: int, tk: int, eps: float, scale: float, gn=True):
    """-> y [rows, 320] (+ y._gn = its GroupNorm partial plane when gn)"""
    c = o1.shape[-1]
    rows = o1.numel() // c
    hw = rows // n_img
    y = torch.empty((rows, c), dtype=o1.dtype, device=o1.device)
    part = torch.empty((n_img, hw // CHAIN_TOK, c, 2), dtype=torch.float32, device=o1.device) if gn else None
    check(lib.ur_transformer_tail_fused(o1.data_ptr(), h0.data_ptr(), xres.data_ptr(), stream_w.data_ptr(), stream_w.numel(), y.data_ptr(),
                                        None if part is None else part.data_ptr(), rows, hw, c, hidden, heads, tk, eps, scale,
                                        ops._dt(o1), ops._stream()))
    if gn:
        y._gn = (part, hw // CHAIN_TOK)
    return y


def ff_geglu_fused(x: torch.Tensor, stream_w: torch.Tensor, hidden: int, eps: float, out=None) -> torch.Tensor:
    """y = x + W2 . GEGLU(W1 . LayerNorm(x) + b1) + b2 in ONE launch; x [..., 320] 16-bit, whole 128-token tiles."""
    c = x.shape[-1]
    rows = x.numel() // c
    out = torch.empty_like(x) if out is None else out
    check(lib.ur_ff_geglu_fused(x.data_ptr(), stream_w.data_ptr(), stream_w.numel(), out.data_ptr(), rows, c, hidden, c, c, eps,
                                ops._dt(x), ops._stream()))
    return out
