"""Generator of the hand-scheduled MFMA phases of the token-stationary chain kernels (unirestore_amd/csrc/tchain_asm.inc).

Each phase is one `asm volatile` block:
  * a fixed list of (accumulator, B fragment, LDS address, immediate offset) MFMA steps whose A fragments (weights) come from
    LDS through a DEPTH-deep rotating set of 128-bit temporaries - the ds_read_b128 of step i + DEPTH is issued right behind the
    MFMA of step i, and every MFMA waits with a COUNTED lgkmcnt for exactly its own fragment.  (hipcc left to itself emits
    `ds_read; s_waitcnt lgkmcnt(0); v_mfma` triples at this register pressure - one wave per SIMD, ~400 live registers - i.e.
    every MFMA eats a full LDS round trip.)
  * optionally the LDS-DMA of a later weight tile (11 x 1 KiB per wave: buffer_load_dwordx4 ... lds, M0 = LDS destination), one
    piece every few MFMAs, so that the ~60 issue cycles a piece costs sit in the shadow of the matrix pipe instead of between
    two phases.  The DMA is invisible to hipcc's s_waitcnt bookkeeping on purpose: the kernel counts vmcnt itself.
Operands stay compiler-allocated ("+a" accumulators, "v" fragments): no hand-owned registers, no clobber lists.

Operand numbering inside a block: accumulators %0.., the temporaries, the B fragments, the LDS address VGPRs, then (DMA blocks)
voffset VGPR, buffer descriptor (4 SGPRs), wave id SGPR; soffset SGPR and LDS base SGPR are "+s" outputs (advanced by the block).
The MFMA mnemonic is a macro argument (bf16 / f16 objects share the schedule).

Run:  python tools/gen_chain_asm.py   (rewrites unirestore_amd/csrc/tchain_asm.inc)
"""
import os

DEPTH = 7
PIECES = 11            # 1-KiB LDS-DMA pieces per wave per tile (4 waves x 11 KiB = one 44-KiB tile)


def dma_lines(j, o_vo, o_rs, o_so, o_ld):
    """instructions that issue piece j (0..PIECES-1) of this wave's share of a tile"""
    out = []
    if j == 0:
        out.append(f"s_mov_b32 m0, %{o_ld}")
    elif j % 4 == 0:                      # the 12-bit instruction offset covers 4 pieces; then both bases move on by 4 KiB
        out += [f"s_add_u32 %{o_so}, %{o_so}, 0x1000", f"s_add_u32 %{o_ld}, %{o_ld}, 0x1000", f"s_mov_b32 m0, %{o_ld}"]
    if j % 4 == 0:
        out.append("s_nop 0")             # SALU write of M0 -> LDS-DMA read of M0
    out.append(f"buffer_load_dwordx4 %{o_vo}, %{o_rs}, %{o_so} offen offset:{(j % 4) * 1024} lds")
    return out


ABL = 0        # timing-only ablation variants of the whole file (main(): tchain_asm_abl{4,5}.inc): 4 = no fragment reads, 5 = no MFMAs


def fmt(name, header, lines, has_mn=True):
    if ABL == 4 and has_mn:
        lines = [ln for ln in lines if not ln.startswith("ds_read") and not ln.startswith("s_waitcnt lgkmcnt")]
    if ABL == 5 and has_mn:
        lines = [ln for ln in lines if not ln.startswith("@MN@")]
    txt = f"// {name}: {header}\n#define {name}{'(MN)' if has_mn else ''} \\\n"
    parts = []
    for ln in lines:
        if ln.startswith("@MN@"):
            parts.append('  MN "' + ln.replace("@MN@", "") + '\\n\\t"')
        else:
            parts.append('  "' + ln + '\\n\\t"')
    return txt + " \\\n".join(parts) + "\n\n"


def block(name, steps, n_acc, n_b, n_addr, pad=True, depth=DEPTH, zero_first=False, dma=False):
    """steps: list of (acc, b, addr, imm).  zero_first: the first MFMA into each accumulator takes 0 as its C operand (the
    accumulator stays a "+a" operand so that hipcc keeps ONE register tuple for it across the whole loop).
    dma: interleave the LDS-DMA of a tile (PIECES pieces)."""
    # asm operand numbers: outputs first (accumulators "+a", temporaries "=&v", [soffset, lds base "+s"]), then inputs
    o_acc, o_tmp = 0, n_acc
    nxt = n_acc + depth
    o_so = o_ld = o_vo = o_rs = None
    if dma:
        o_so, o_ld = nxt, nxt + 1
        nxt += 2
    o_b, o_addr = nxt, nxt + n_b
    nxt += n_b + n_addr
    o_wid = None
    if dma:
        o_vo, o_rs, o_wid = nxt, nxt + 1, nxt + 2
        nxt += 3
    assert nxt <= 30, (name, nxt)
    n = len(steps)
    lines = ["s_waitcnt lgkmcnt(0)"]

    def rd(i):
        a, b, ad, imm = steps[i]
        return f"ds_read_b128 %{o_tmp + i % depth}, %{o_addr + ad} offset:{imm}"

    for i in range(min(depth, n)):
        lines.append(rd(i))
    # DMA: wave w of the workgroup issues ALL its 11 pieces in one burst behind MFMA number ~ (w + 1) * n / 4 (runtime branch on
    # the wave id): the four waves share ONE texture-address path per CU (64 B/clk: 16+ cycles per 1-KiB piece), so pieces issued
    # by all four at the same point of the phase queue up and each costs its wave ~80 cycles of issue; staggered, a wave stalls for
    # its own ~200 cycles while the other three SIMDs keep multiplying.
    dma_at = {}
    if dma:
        for w in range(4):
            dma_at[min(n - 1, ((w + 1) * n) // 4 - 1)] = w
    seen = set()
    for i, (a, b, ad, imm) in enumerate(steps):
        issued = min(n, i + depth)
        lines.append(f"s_waitcnt lgkmcnt({issued - (i + 1)})")
        srcc = "0" if (zero_first and a not in seen) else f"%{o_acc + a}"
        seen.add(a)
        lines.append(f"@MN@ %{o_acc + a}, %{o_tmp + i % depth}, %{o_b + b}, {srcc}")
        if i + depth < n:
            lines.append(rd(i + depth))
        if i in dma_at:
            w = dma_at[i]
            lines += [f"s_cmp_lg_u32 %{o_wid}, {w}", f"s_cbranch_scc1 .Ltc%=_{w}"]
            for j in range(PIECES):
                lines += dma_lines(j, o_vo, o_rs, o_so, o_ld)
            lines.append(f".Ltc%=_{w}:")
    if pad:
        lines.append("s_nop 15")          # MFMA result -> VALU / accvgpr read of the same registers (software-managed hazard)
    hdr = f"{n} MFMAs, {n_acc} accumulators (%0..), {depth} temporaries (%{o_tmp}..), {n_b} B fragments (%{o_b}..), {n_addr} addresses (%{o_addr}..)"
    if dma:
        hdr += f", DMA: soffset %{o_so}, lds base %{o_ld}, voffset %{o_vo}, descriptor %{o_rs}, wave id %{o_wid}"
    return fmt(name, hdr, lines)


def dma_block(name):
    """stand-alone LDS-DMA of one wave's share of a tile: outputs soffset %0, lds base %1 ("+s"); inputs voffset %2, descriptor %3"""
    lines = []
    for j in range(PIECES):
        lines += dma_lines(j, 2, 3, 0, 1)
    return fmt(name, f"{PIECES} LDS-DMA pieces: soffset %0, lds base %1, voffset %2, descriptor %3", lines, has_mn=False)


def aux_block(name, offsets):
    """fp32 vectors of a tile's aux area -> registers: len(offsets) ds_read_b128 at (address operand %n) + (constant operand %n+1)
    + immediate, waited for inside the block (outputs are early-clobber: the statement is complete when it ends).  The
    fragment-dependent part of the offset is a compile-time constant OPERAND: one address register (aux base + lane half) serves
    every fragment of a tile - hipcc otherwise keeps one address per fragment alive and spills them.  hipcc must not see these
    reads: a compiler-visible ds_read behind an LDS-DMA makes it drain the whole DMA queue (s_waitcnt vmcnt(0)) first."""
    n = len(offsets)
    lines = [f"ds_read_b128 %{i}, %{n} offset:%c{n + 1}+{off}" for i, off in enumerate(offsets)] + ["s_waitcnt lgkmcnt(0)"]
    return fmt(name, f"{n} x 16 bytes of the aux area (%0..%{n - 1}), address %{n}, constant byte offset %{n + 1}", lines, has_mn=False)


def gemm_ktile(nf, ks=4):
    """one 64-deep k tile of an N = 32*nf stage: block [32*nf rows][128 B]; address operand s = k-step, fragment f at f*4096"""
    return [(f, s, s, f * 4096) for s in range(ks) for f in range(nf)]


def ff1(kb0, nkb):
    """GEGLU up-projection, blocks kb0 .. kb0+nkb-1 of [a rows | g rows][128 B] (8 KiB per block); B fragment index is local"""
    return [(ag, (kb - kb0) * 4 + s, s, kb * 8192 + ag * 4096) for kb in range(kb0, kb0 + nkb) for s in range(4) for ag in range(2)]


def main():
    global ABL
    here = os.path.dirname(os.path.abspath(__file__))
    for ABL in (4, 5, 0):
        write(os.path.join(here, "ab", f"tchain_asm_abl{ABL}.inc") if ABL else os.path.join(here, "..", "unirestore_amd", "csrc", "tchain_asm.inc"))


def write(out):
    txt = "// GENERATED by tools/gen_chain_asm.py - do not edit (hand-scheduled MFMA phases of tchain.hip; see the generator's docstring)\n#pragma once\n\n"
    txt += dma_block("TC_ASM_DMA")
    txt += block("TC_ASM_GEMM_N10", gemm_ktile(10), 10, 4, 4, dma=True)
    txt += block("TC_ASM_GEMM_N10_Z", gemm_ktile(10), 10, 4, 4, zero_first=True, dma=True)
    txt += block("TC_ASM_FF1_A", ff1(0, 3), 2, 12, 4, pad=False, zero_first=True, dma=True)
    txt += block("TC_ASM_FF1_B", ff1(3, 2), 2, 8, 4)
    # cross-attention, one head of 64: S^T = K q^T over 3 fragments of 32 keys; O^T = V^T P^T over 5 k-steps of 16 keys
    txt += block("TC_ASM_ATT_S", gemm_ktile(3), 3, 4, 4, zero_first=True, dma=True)
    txt += block("TC_ASM_ATT_PV", [(f, s, s, f * 4096) for s in range(5) for f in range(2)], 2, 5, 5, zero_first=True)
    # FF1 epilogue vectors: (bias a | bias g | colsum a | colsum g) x 32 floats; lane half h is folded into the address operand
    # (one half-fragment u per block: 8 x float4 = 32 registers live at a time)
    txt += aux_block("TC_ASM_AUX_FF1", [vec * 4 + 16 * q for vec in (0, 32, 64, 96) for q in range(2)])
    # a stage's bias for 4 half-fragments = 2 accumulator fragments: 8 floats each (32 registers live at a time)
    txt += aux_block("TC_ASM_AUX_8", [u * 64 + q * 16 for u in range(4) for q in range(2)])
    # ONE fragment of a LayerNorm-folded stage: bias (floats 0..) for its two half-fragments, then column sums (floats 512..)
    txt += aux_block("TC_ASM_AUX_BC8", [base + u * 64 + q * 16 for base in (0, 2048) for u in range(2) for q in range(2)])
    open(out, "w").write(txt)
    print("wrote", os.path.normpath(out))


if __name__ == "__main__":
    main()
