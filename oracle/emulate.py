"""Rounding-emulation harness for the error budget (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Runs the fp32 oracle graph with 16-bit rounding injected at selectable points, so that the distance between the HIP
path and the fp32 oracle can be split into
  * "operand":  every contraction (conv / linear / Q.K^T / P.V) sees its two operands rounded to the 16-bit type and
                accumulates in fp32 - the floor of ANY implementation that feeds the matrix cores 16-bit operands;
  * "storage":  additionally every activation tensor that the HIP path keeps in HBM (conv / linear / norm outputs,
                residual sums) is rounded once more - what storing activations as 16-bit adds on top.
Used by tests/test_error_budget.py and tools/error_budget.py; DESIGN.md section 4 quotes the numbers.
"""
import contextlib

# The budget itself: rel-L2 (z0, zt, image) of the fp32 oracle with "operands + stored activations" rounding against the plain fp32
# oracle, FULL-SIZE architecture (sd-turbo widths, seeded random weights), 256x256 input, 1 DDIM step - the output of
# `python tools/error_budget.py --full --size 256 --steps 1` (a CPU run of several minutes; DESIGN.md section 4 quotes the table).
# The full-size GPU parity tests take their bf16 tolerance from this row (x BUDGET_MARGIN), not from what the HIP path measured.
BUDGET_FULLSIZE = {"bf16": dict(operands=(3.45e-3, 3.07e-3, 3.11e-3), storage=(5.83e-3, 4.61e-3, 4.69e-3)),
                   "fp16": dict(operands=(4.3e-4, 3.7e-4, 3.9e-4), storage=(7.4e-4, 5.6e-4, 5.8e-4))}
BUDGET_MARGIN = 1.25          # sample-to-sample spread of the emulation itself (different image / noise / resolution)

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import blocks


def _rnd(t, dt):
    return t.to(dt).float() if t is not None and t.is_floating_point() else t


@contextlib.contextmanager
def rounding(model: nn.Module, dtype=torch.bfloat16, operands=True, storage=False):
    """Context manager: patch the contractions used by the oracle graph (F.conv2d, F.linear, the two matmuls of
    blocks._sdpa) and, for storage=True, hook every Conv2d / Linear / GroupNorm / LayerNorm / residual-block output."""
    conv0, lin0, sdpa0 = F.conv2d, F.linear, blocks._sdpa

    def conv(x, w, b=None, *a, **k):
        return conv0(_rnd(x, dtype), _rnd(w, dtype), b, *a, **k)

    def lin(x, w, b=None):
        return lin0(_rnd(x, dtype), _rnd(w, dtype), b)

    def sdpa(q, k, v, heads):
        import math
        b, tq, c = q.shape
        d = c // heads
        q, k, v = (_rnd(t, dtype) for t in (q, k, v))
        q = q.view(b, tq, heads, d).transpose(1, 2)
        k = k.view(b, -1, heads, d).transpose(1, 2)
        v = v.view(b, -1, heads, d).transpose(1, 2)
        w = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(d), dim=-1)
        return (_rnd(w, dtype) @ v).transpose(1, 2).reshape(b, tq, c)

    handles = []
    if storage:
        leaf = (nn.Conv2d, nn.Linear, nn.GroupNorm, nn.LayerNorm, nn.InstanceNorm2d)
        blk = (blocks.ResnetBlock2D, blocks.Transformer2DModel, blocks.BasicTransformerBlock, blocks.AttentionBlock)

        def hook(_m, _i, out):
            return _rnd(out, dtype) if torch.is_tensor(out) else out

        for m in model.modules():
            if isinstance(m, leaf) or isinstance(m, blk) or type(m).__name__ in ("CSCEAdapter", "NAFBlock", "AdaNAFV2"):
                handles.append(m.register_forward_hook(hook))
    if operands:
        F.conv2d, F.linear, blocks._sdpa = conv, lin, sdpa
    try:
        yield
    finally:
        F.conv2d, F.linear, blocks._sdpa = conv0, lin0, sdpa0
        for h in handles:
            h.remove()


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())
