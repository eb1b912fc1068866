"""Generator of the hand-scheduled K-tile bodies of the implicit-GEMM kernels (unirestore_amd/csrc/igemm_asm.inc).

One macro = the MFMAs of ONE 64-deep K tile of a wave whose output tile is FM x FN fragments of 32 x 32 (FM activation
fragments = B operands, FN weight fragments = A operands), both operand kinds read from LDS rows of 128 bytes whose eight
16-byte slots are XOR-swizzled with (row >> 1) & 7:

    k-step ks (0..3), lane half h:  slot = (2 ks + h) ^ swz   ->   address(ks) = address(0) ^ (ks << 5)

so the caller passes ONE address per activation fragment and ONE for the weight fragments (fragment a of the wave sits
a * 4096 bytes further: 32 rows) and the block derives the other k-steps with one v_xor each.

Why asm: hipcc schedules this loop as `ds_read ...; s_waitcnt lgkmcnt(0); v_mfma` groups - every one or two MFMAs eat a full
LDS round trip (measured: 40 % MFMA-busy on the dominant conv3x3 launch, profiles/r3_b_pmc_halo_conv.txt) - and keeps one
address register per (tap, k-step, fragment) alive (36-72 VGPRs).  Here the fragment reads run POOL deep ahead of the MFMAs
through a rotating register pool and every MFMA waits with a COUNTED lgkmcnt for exactly its own operands.

Operands: %0.. accumulators ("+v", index a * FM + b), then POOL temporaries, FM + 1 address temporaries ("=&v"), then inputs:
FM activation addresses, 1 weight address ("v").  The MFMA mnemonic is a macro argument.
"""
import os

import gen_chain_asm
from gen_chain_asm import fmt


def ktile(name, fm, fn, pool, pad=False):
    n_acc = fm * fn
    o_acc, o_tmp = 0, n_acc
    o_xb = o_tmp + pool                  # fm xor temporaries for the activation addresses
    o_xw = o_xb + fm                     # 1 for the weight address
    o_ab = o_xw + 1                      # inputs: fm activation addresses
    o_aw = o_ab + fm                     # weight address
    assert o_aw + 1 <= 30, name
    # reads in order of first use; MFMA order: k-step major, then a, then b
    mfmas = [(ks, a, b) for ks in range(4) for a in range(fn) for b in range(fm)]
    reads, first, last = [], {}, {}
    for i, (ks, a, b) in enumerate(mfmas):
        for key in (("B", ks, b), ("A", ks, a)):
            if key not in first:
                first[key] = i
                reads.append(key)
            last[key] = i
    lines = ["s_waitcnt lgkmcnt(0)"]
    free = list(range(pool))
    where = {}                           # read key -> pool register
    issued = []                          # keys in issue order
    xor_done = set()

    def issue():
        key = reads[len(issued)]
        kind, ks, idx = key
        if ks > 0 and (kind, ks, idx if kind == "B" else 0) not in xor_done:
            xor_done.add((kind, ks, idx if kind == "B" else 0))
            dst, src = (o_xb + idx, o_ab + idx) if kind == "B" else (o_xw, o_aw)
            lines.append(f"v_xor_b32 %{dst}, {hex(ks << 5)}, %{src}")
        if kind == "B":
            addr = (o_ab + idx) if ks == 0 else (o_xb + idx)
            off = 0
        else:
            addr = o_aw if ks == 0 else o_xw
            off = idx * 4096
        r = free.pop(0)
        where[key] = r
        lines.append(f"ds_read_b128 %{o_tmp + r}, %{addr}" + (f" offset:{off}" if off else ""))
        issued.append(key)

    def can_issue():
        return len(issued) < len(reads) and free

    while can_issue():
        issue()
    for i, (ks, a, b) in enumerate(mfmas):
        ka, kb = ("A", ks, a), ("B", ks, b)
        while ka not in where or kb not in where:       # (pool too small to hold this MFMA's operands: cannot happen for pool >= 2)
            issue()
        need = max(issued.index(ka), issued.index(kb))
        lines.append(f"s_waitcnt lgkmcnt({len(issued) - 1 - need})")
        lines.append(f"@MN@ %{o_acc + a * fm + b}, %{o_tmp + where[ka]}, %{o_tmp + where[kb]}, %{o_acc + a * fm + b}")
        for key in (ka, kb):
            if last[key] == i:
                free.append(where[key])
        n = 0
        while can_issue() and n < 2:
            issue()
            n += 1
    if pad:
        lines += ["s_nop 15", "s_nop 7"]
    hdr = (f"{len(mfmas)} MFMAs / {len(reads)} fragment reads; accumulators %0..%{n_acc - 1} (a * {fm} + b), pool %{o_tmp}..%{o_xb - 1}, "
           f"address temporaries %{o_xb}..%{o_xw}, activation addresses %{o_ab}.., weight address %{o_aw}")
    return fmt(name, hdr, lines)


def ktile_pipe(name, fm, fn, pa_n, pb_n, split):
    """Tap-crossing software pipeline (halo convs): the fragment reads of a K tile's first k-step are issued by the PREVIOUS tile's
    block, so the matrix pipe does not drain at the per-tile barrier.  One K tile = two asm blocks with the workgroup's
    "next weight tile has landed" wait + barrier between them (the caller's C++):
        NAME_PRE   reads of k-step 0 of the first tile (addresses = the "next" operands)
        NAME_H1    MFMAs 0 .. split-1
        NAME_H2    MFMAs split .. end + reads of k-step 0 of the next tile ("next" addresses; its weights were published by the
                   barrier in front of this block)
        NAME_H2L   the same without the prefetch (last tile)
    Fragment registers are two rings that persist ACROSS blocks ("+v" operands): weight read j of a tile lives in A[j % pa_n],
    activation read j in B[j % pb_n] (reads per tile are multiples of the ring sizes, so the mapping is the same for every tile).
    lgkmcnt is counted over the whole stream: at block entry exactly fm + fn prefetch reads are outstanding (LDS ops the compiler
    issued in between are younger and only make the counted waits stricter).
    Operands (identical lists for all four): accumulators, pa_n + pb_n ring registers ("+v"), fm + 1 address temporaries ("=&v"),
    then inputs: fm activation addresses + weight address of THIS tile, the same of the NEXT tile."""
    n_acc = fm * fn
    assert (4 * fn) % pa_n == 0 and (4 * fm) % pb_n == 0
    o_acc, o_pa = 0, n_acc
    o_pb = o_pa + pa_n
    o_xb = o_pb + pb_n
    o_xw = o_xb + fm
    o_ab = o_xw + 1
    o_aw = o_ab + fm
    o_abn = o_aw + 1
    o_awn = o_abn + fm
    assert o_awn + 1 <= 30, name
    mfmas = [(ks, a, b) for ks in range(4) for a in range(fn) for b in range(fm)]
    reads, last = [], {}
    for i, (ks, a, b) in enumerate(mfmas):
        for key in (("B", ks, b), ("A", ks, a)):
            if key not in last:
                reads.append(key)
            last[key] = i
    npre = fm + fn
    assert all(k[1] == 0 for k in reads[:npre])

    def reg(key):
        kind, ks, idx = key
        return (o_pb + (ks * fm + idx) % pb_n) if kind == "B" else (o_pa + (ks * fn + idx) % pa_n)

    def read_line(key, nxt, xor_done, lines):
        kind, ks, idx = key
        base_b, base_w = (o_abn, o_awn) if nxt else (o_ab, o_aw)
        if ks > 0:
            tag = (kind, ks, idx if kind == "B" else 0)
            if tag not in xor_done:
                xor_done.add(tag)
                dst, src = (o_xb + idx, base_b + idx) if kind == "B" else (o_xw, base_w)
                lines.append(f"v_xor_b32 %{dst}, {hex(ks << 5)}, %{src}")
        if kind == "B":
            addr, off = ((base_b + idx) if ks == 0 else (o_xb + idx)), 0
        else:
            addr, off = (base_w if ks == 0 else o_xw), idx * 4096
        lines.append(f"ds_read_b128 %{reg(key)}, %{addr}" + (f" offset:{off}" if off else ""))

    out = ""
    # PRE
    lines = []
    for key in reads[:npre]:
        read_line(key, True, set(), lines)
    out += fmt(name + "_PRE", f"reads of k-step 0 of the first tile ({npre}), next-tile address operands", lines)

    for variant in ("", "L"):
        # simulate one steady-state tile; seq numbers: prefetched reads 0..npre-1
        issued = list(reads[:npre])                    # this tile's reads issued so far (stream order)
        total = npre                                   # reads issued in the whole stream up to now (for lgkmcnt)
        seq = {k: i for i, k in enumerate(issued)}
        occupant_dead_at = {}                          # ring register -> MFMA index after which it is free (-1 = free)
        for k in issued:
            occupant_dead_at[reg(k)] = last[k]
        nxt_issued = 0
        halves = {1: [], 2: []}
        xor_done = {1: set(), 2: set()}

        def try_issue(i_done, half, limit):
            """issue up to `limit` reads of the stream whose ring register is free after MFMA i_done"""
            nonlocal total, nxt_issued
            n = 0
            while n < limit:
                if len(issued) < len(reads):
                    key, nxt = reads[len(issued)], False
                elif half == 2 and variant == "" and nxt_issued < npre:
                    key, nxt = reads[nxt_issued], True
                else:
                    return
                r = reg(key)
                if occupant_dead_at.get(r, -1) > i_done:
                    return
                read_line(key, nxt, xor_done[half] if not nxt else set(), halves[half])
                if nxt:
                    nxt_issued += 1
                    occupant_dead_at[r] = 10 ** 9      # lives into the next tile
                else:
                    issued.append(key)
                    seq[key] = total
                    occupant_dead_at[r] = last[key]
                total += 1
                n += 1

        for i, (ks, a, b) in enumerate(mfmas):
            half = 1 if i < split else 2
            ka, kb = ("A", ks, a), ("B", ks, b)
            while ka not in seq or kb not in seq:
                before = total
                try_issue(i - 1, half, 1)
                assert total > before, (name, "ring too small", i)
            need = max(seq[ka], seq[kb])
            halves[half].append(f"s_waitcnt lgkmcnt({total - 1 - need})")
            halves[half].append(f"@MN@ %{o_acc + a * fm + b}, %{reg(ka)}, %{reg(kb)}, %{o_acc + a * fm + b}")
            try_issue(i, half, 2)
        try_issue(len(mfmas) - 1, 2, 99)               # whatever prefetch is left
        if variant == "":
            assert nxt_issued == npre and len(issued) == len(reads), name
            out += fmt(name + "_H1", f"MFMAs 0..{split - 1} of a tile ({npre} prefetched reads outstanding at entry)", halves[1])
            out += fmt(name + "_H2", f"MFMAs {split}..{len(mfmas) - 1} + the {npre} k-step-0 reads of the next tile", halves[2])
        else:
            out += fmt(name + "_H2L", f"MFMAs {split}..{len(mfmas) - 1} of the last tile (nothing outstanding at exit)", halves[2] + ["s_nop 15", "s_nop 7"])
    return out


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    for abl in (4, 5, 0):             # timing-only variants (4 = no fragment reads, 5 = no MFMAs) for A/B builds, then the real file
        gen_chain_asm.ABL = abl
        # the shipped file lives in csrc/; the timing-only variants next to the A/B tooling (tools/ab/, selected by -DUR_IGASM_ABL=4|5)
        write(os.path.join(here, "ab", f"igemm_asm_abl{abl}.inc") if abl else os.path.join(here, "..", "unirestore_amd", "csrc", "igemm_asm.inc"))


def write(out):
    txt = "// GENERATED by tools/gen_igemm_asm.py - do not edit (hand-scheduled K-tile bodies of igemm_impl.h; see the generator's docstring)\n#pragma once\n\n"
    txt += ktile("IG_ASM_KT_1x5", 1, 5, 10)
    txt += ktile("IG_ASM_KT_2x2", 2, 2, 8)
    txt += ktile("IG_ASM_KT_1x2", 1, 2, 6)
    txt += ktile("IG_ASM_KT_2x1", 2, 1, 6)
    txt += ktile("IG_ASM_KT_1x1", 1, 1, 6)
    txt += ktile("IG_ASM_KT_2x4", 2, 4, 6)
    txt += ktile("IG_ASM_KT_4x2", 4, 2, 8)
    txt += ktile("IG_ASM_KT_1x4", 1, 4, 8)
    txt += ktile("IG_ASM_KT_4x1", 4, 1, 8)
    txt += ktile("IG_ASM_KT_1x10", 1, 10, 6)
    txt += ktile_pipe("IG_ASM_KP_1x5", 1, 5, 10, 4, 10)
    txt += ktile_pipe("IG_ASM_KP_2x2", 2, 2, 4, 4, 8)
    open(out, "w").write(txt)
    print("wrote", os.path.normpath(out))


if __name__ == "__main__":
    main()
