"""Generator of the hand-scheduled main loop of the ping-pong attention kernel (unirestore_amd/csrc/attention_pp_asm.inc).

The whole tile loop + epilogue of attn_pp64_kernel (attention_pp.hip) is ONE `asm volatile` statement with hand-owned registers
(the clobber list keeps hipcc out of v[LO:255]); hipcc only computes the addresses / descriptors that enter it.  Why: left to
itself hipcc (a) schedules the MFMA phase as `ds_read_b128; s_waitcnt lgkmcnt(0); v_mfma` triples - every MFMA eats an LDS round
trip -, (b) merges the fast / slow softmax paths with 48 v_mov copies, (c) SLP-packs the row sums into v_pk_add_f32 (slower
than two v_add_f32 beside MFMAs) and (d) sinks the P packing behind the phase barrier.  (First build of this kernel, compiler
scheduled: 179.5 us on the 4 x 4096 x 4096 x 64 shape against 173.4 for the round-1 kernel.)

Structure (one workgroup = 8 waves = groups A (waves 0-3) and B (waves 4-7), one wave of each per SIMD, 32 queries per wave):

    between barrier #t and #t+1   A: P.V(t) + K.Q^T(t+1) interleaved + DMA of stage t+3, then softmax(t+1)
                                  B: softmax(t), then P.V(t) + K.Q^T(t+1) + DMA of stage t+3
    ONE s_barrier per tile (body_onebar has the LDS-safety argument); a wave waits for its own older DMA pieces (vmcnt(2)) at the
    end of its softmax phase.  (The first version had a barrier after every phase: ATTN_ONEBAR=0, and the ATTN_DBG=1 timer build.)

  stage s = { V^T(s) | K(s+1) } (16 KiB) lives in LDS buffer s & 3; the loop is unrolled by four so every LDS offset is an
  immediate.  MFMA phase: P.V and K.Q^T alternate (four independent accumulator chains), fragments through an 8-slot register
  ring with counted lgkmcnt, the first four reads issued in front of the phase barrier, the two DMA pieces of stage t+3 behind
  MFMAs 3 and 9.  Softmax fast path per 32 scores: 32 v_exp_f32 + 32 v_add_f32 + 16 v_cvt_pk + 4; the S accumulators start at -m
  (the MFMA's C operand), so there is no subtraction and no maximum: the lane's partial row sum proves p < 2^14, else the slow path
  (maximum, new reference, O / l rescale, recompute) runs - always at tile 0.

Measured on the way (4 x 4096 x 4096 x 64 launch = 2 full rounds of workgroups, us; round-1 kernel 173.4): two LDS buffers, DMA
in the phases, all fragment reads behind the barrier 146.4; four buffers + prefetch 141.8; interleaved chains 143.3 (no change:
the chains were not the limit); v_pk_add_f32 row sums 154.4, v_dot2c 158.7 (both slower than scalar adds); DMA placement (softmax
phase start / inside the exponentials / behind MFMAs) 142.2 / 145.9 / 142.1.  In-kernel timers (ATTN_DBG=1): softmax phase ~840
cycles (707 without the partner's MFMAs: 32 x ~10 for the exponentials + 64 x 4 + overhead), MFMA phase ~740 (512 of MFMA), clock
1.87 GHz; without any barrier the same work takes 1600 instead of 2012 cycles per tile - and the clock drops to 1.67 GHz.

Operand numbering: see OPERANDS below (outputs = SGPR temporaries first, then the inputs).  Macro arguments: MFMA mnemonic,
16-bit pack mnemonic (bf16 / f16 objects share the schedule).

Run:  python tools/gen_attn_asm.py   (rewrites unirestore_amd/csrc/attention_pp_asm.inc)
"""
import os

# ---------------------------------------------------------------------------------------------------------- operands
OUT_S = ["t", "first", "last", "vso", "kso"]                                                        # "=&s" temporaries
IN_V = ["qf0", "qf1", "qf2", "qf3", "ka0", "ka1", "va0", "va1", "kvo", "vvo", "optr"]  # "v" inputs
IN_S = ["rsk", "rsv", "kstep", "nt", "grp", "ldsb", "part"]                             # "s" inputs
DBG = int(os.environ.get("ATTN_DBG", "0"))      # phase timers (tools/attn_phase_timers.py): 5 more SGPR temporaries, s_memtime into VCC
if DBG:
    OUT_S += ["acc0", "acc1", "acc2", "acc3", "tprev", "cyc0", "rt0"]
OPERANDS = OUT_S + IN_V + IN_S
OP = {n: f"%{i}" for i, n in enumerate(OPERANDS)}

# ---------------------------------------------------------------------------------------------------------- hand-owned VGPRs
_next = [256]


def alloc(n, align=4):
    base = (_next[0] - n) // align * align
    _next[0] = base
    return base


S = alloc(32)          # S^T - m : fragment f at S + 16 f
O = alloc(32)          # O^T accumulators: fragment f at O + 16 f
NEGM = alloc(16)       # -m in all 16 registers (C operand of the first K.Q^T MFMA of a fragment)
RING = alloc(32)       # 8 x 128-bit fragment slots
P = alloc(16)          # packed P^T: key step kk at P + 4 kk
KA = alloc(8, 1)       # K fragment addresses: KA + 4 f + ds
VA = alloc(8, 1)       # V^T fragment addresses: VA + 4 f + kk
TMP = alloc(8)
E = alloc(32) if int(os.environ.get("ATTN_LATE_CVT", "0")) else None     # fp32 exp2 results kept for the late packing (LATE_CVT)
PS0 = alloc(2, 2)
PS1 = PS0 + 1
PSUM, LRUN, MC, MX, DLT, ALPHA = (alloc(1, 1) for _ in range(6))
LO = _next[0]


def v(i, n=1):
    return f"v{i}" if n == 1 else f"v[{i}:{i + n - 1}]"


# row sums: "add" = two v_add_f32 per score pair; "pk" = one v_pk_add_f32 (measured 11 % SLOWER on the whole kernel: the packed fp32
# add issues far below the scalar rate); "dot2" = one v_dot2c of the PACKED pair against (1, 1): the normaliser then sums exactly
# the 16-bit weights the P.V MFMA multiplies
SUM_MODE = os.environ.get("ATTN_SUM", "add")
PK_ADD = SUM_MODE == "pk"
# ATTN_LATE_CVT=1: the 16-bit packing of P for key steps 1-3 (12 of the 16 v_cvt_pk) moves from the softmax phase - the longer of
# the two - into the wave's own MFMA phase, four packs in front of each key step's first P.V MFMA.  Measured: no gain (143.4 vs
# 143.0 us; the softmax phase is bound by the 32 v_exp_f32 at ~10 cycles each, not by its instruction count) and it needs an
# s_nop between the pack and the MFMA that reads it (VALU write -> MFMA source: a real hazard, 16 red tests without it). Off.
LATE_CVT = int(os.environ.get("ATTN_LATE_CVT", "0"))
OUT_WT = int(os.environ.get("ATTN_OUT_WT", "0"))         # write-through stores of the output rows: +2.3 ms per forward (plain: the TAIL
                                                         # chain re-reads them at once; common.h store16_wt has the A/B table)
TILE = 8192
STAGE = 16384
SUM_MAX = 0x46800000      # 2^14


class Prog:
    def __init__(self):
        self.lines = []

    def __call__(self, s):
        self.lines.append(s)

    def label(self, name):
        self.lines.append(f".Lap%=_{name}:")

    @staticmethod
    def L(name):
        return f".Lap%=_{name}"


NB = 4                    # LDS stage buffers (the loop is unrolled NB times so every LDS offset is an immediate)


def dma_piece(p, buf, which):
    """one 1-KiB piece of stage s -> buffer buf: which = "v": V^T(s), "k": K(s+1)"""
    if which == "v":
        p(f"s_add_u32 m0, {OP['ldsb']}, {hex(buf * STAGE)}")
        p("s_nop 0")
        p(f"buffer_load_dwordx4 {OP['vvo']}, {OP['rsv']}, {OP['vso']} offen lds")
        p(f"s_add_u32 {OP['vso']}, {OP['vso']}, 0x80")
    else:
        p(f"s_add_u32 m0, {OP['ldsb']}, {hex(buf * STAGE + TILE)}")
        p("s_nop 0")
        p(f"buffer_load_dwordx4 {OP['kvo']}, {OP['rsk']}, {OP['kso']} offen lds")
        p(f"s_add_u32 {OP['kso']}, {OP['kso']}, {OP['kstep']}")


PV = [(kk, f) for kk in range(4) for f in range(2)]        # MFMA order of P.V: key step major (the two accumulators alternate)
QK = [(ds, f) for ds in range(4) for f in range(2)]        # and of K.Q^T


def qk_only(p, buf):
    """K.Q^T of one tile from stage buffer `buf` (prologue form: nothing else in flight)"""
    off = buf * STAGE + TILE
    for j, (ds, f) in enumerate(QK):
        p(f"ds_read_b128 {v(RING + 4 * j, 4)}, {v(KA + 4 * f + ds)} offset:{off}")
    for j, (ds, f) in enumerate(QK):
        p(f"s_waitcnt lgkmcnt({7 - j})")
        c = v(NEGM, 16) if ds == 0 else v(S + 16 * f, 16)
        p(f"@MN@ {v(S + 16 * f, 16)}, {v(RING + 4 * j, 4)}, {OP['qf%d' % ds]}, {c}")
    p("s_nop 15")


NRING = 8       # fragment ring slots
WAIT_EVERY = int(os.environ.get("ATTN_WAIT_EVERY", "1"))     # MFMAs per counted lgkmcnt wait (2 / 4: +-0 - as is dropping s_setprio: the launch
                                                            # sits at the socket power cap, re-ordering instructions does not move it)
NPRE = int(os.environ.get("ATTN_NPRE", "4"))        # fragment reads issued in front of the phase barrier (the rest of the ring behind it)
DMA_AT = (3, 9)  # the phase's two DMA pieces go behind these MFMAs (issue slots in the shadow of the matrix pipe)


def mfma_seq(with_qk):
    """MFMA order of a phase: P.V(t) and K.Q^T(t+1) interleaved, so that four independent accumulator chains (O0, S0, O1, S1)
    alternate."""
    if not with_qk:
        return [("pv", j) for j in range(8)]
    return [x for j in range(8) for x in (("pv", j), ("qk", j))]


def frag_read(p, kind, j, slot, buf):
    if kind == "pv":
        kk, f = PV[j]
        p(f"ds_read_b128 {v(RING + 4 * slot, 4)}, {v(VA + 4 * f + kk)} offset:{buf * STAGE}")
    else:
        ds, f = QK[j]
        p(f"ds_read_b128 {v(RING + 4 * slot, 4)}, {v(KA + 4 * f + ds)} offset:{buf * STAGE + TILE}")


def prefetch(p, buf, with_qk):
    """the first NPRE fragment reads of the coming MFMA phase, issued in front of the phase barrier (the stage is complete since
    the barrier before this softmax phase; the ring is idle during a softmax phase)"""
    for i, (kind, j) in enumerate(mfma_seq(with_qk)[:NPRE]):
        frag_read(p, kind, j, i % NRING, buf)


def mfma_phase(p, buf, with_qk, dma=None):
    """P.V(t) and K.Q^T(t+1), both from stage buffer `buf`.  NPRE fragments are already in flight (prefetch), the rest of the ring
    is issued first thing; each MFMA frees its ring slot for the read NRING places further on and waits with a counted lgkmcnt for
    exactly its own fragment.  dma = (buffer, [(piece, label to skip to or None)]): the LDS-DMA of a later stage rides behind
    MFMAs DMA_AT."""
    seq = mfma_seq(with_qk)
    n = len(seq)
    for i, (kind, j) in enumerate(seq[NPRE:NRING]):
        frag_read(p, kind, j, (NPRE + i) % NRING, buf)
    pieces = list(dma[1]) if dma else []
    for i, (kind, j) in enumerate(seq):
        issued = min(n, NRING + i)
        if i % WAIT_EVERY == 0:          # one counted wait covers the fragments of the next WAIT_EVERY MFMAs (all issued already)
            p(f"s_waitcnt lgkmcnt({issued - (min(i + WAIT_EVERY, n))})")
        slot = v(RING + 4 * (i % NRING), 4)
        if kind == "pv" and LATE_CVT and PV[j][0] >= 1 and PV[j][1] == 0:        # first use of P for this key step: pack it now
            for u in range(4):
                q = 4 * PV[j][0] + u
                p(f"@CVT@ {v(P + q)}, {v(E + 2 * q)}, {v(E + 2 * q + 1)}")
            p("s_nop 1")           # VALU write -> MFMA source read
        if kind == "pv":
            kk, f = PV[j]
            p(f"@MN@ {v(O + 16 * f, 16)}, {slot}, {v(P + 4 * kk, 4)}, {v(O + 16 * f, 16)}")
        else:
            ds, f = QK[j]
            c = v(NEGM, 16) if ds == 0 else v(S + 16 * f, 16)
            p(f"@MN@ {v(S + 16 * f, 16)}, {slot}, {OP['qf%d' % ds]}, {c}")
        if i + NRING < n:
            kind2, j2 = seq[i + NRING]
            frag_read(p, kind2, j2, i % NRING, buf)
        if i in DMA_AT and pieces:
            which, cond_label = pieces.pop(0)
            if cond_label:                   # no such tile in the final trip
                p(f"s_cmp_eq_u32 {OP['last']}, 1")
                p(f"s_cbranch_scc1 {Prog.L(cond_label)}")
            dma_piece(p, dma[0], which)
            if cond_label:
                p.label(cond_label)
    p("s_nop 7")           # MFMA result -> VALU read of the same registers (software-managed hazard; the barrier adds the rest)


def exp_sum_pack(p, src_sub=None, hooks=None):
    """P = pack(exp2(S [- DLT])), PS0 / PS1 = the two interleaved partial row sums.  Software-pipelined over the 16 register pairs:
    the exponentials of pair i + 1 are issued before the adds / pack of pair i (a transcendental's result must not be read by the
    next instruction)."""
    p(f"v_mov_b32 {v(PS0)}, 0")
    p(f"v_mov_b32 {v(PS1)}, 0")
    assert PS1 == PS0 + 1 and PS0 % 2 == 0 and TMP % 2 == 0

    def tmp(i, h):
        return (E + 2 * i + h) if LATE_CVT else (TMP + 2 * (i % 4) + h)

    def exps(i):
        for h in range(2):
            src = v(S + 2 * i + h)
            dst = v(tmp(i, h))
            if src_sub is not None:
                p(f"v_sub_f32 {dst}, {src}, {v(src_sub)}")
                src = dst
            p(f"@EXP@ {dst}, {src}")

    def fin(i):
        t0, t1 = tmp(i, 0), tmp(i, 1)
        if PK_ADD:
            p(f"v_pk_add_f32 {v(PS0, 2)}, {v(PS0, 2)}, {v(t0, 2)}")
        elif SUM_MODE == "dot2":
            p(f"@CVT@ {v(P + i)}, {v(t0)}, {v(t1)}")
            p(f"@DOT@ {v(PS0 + (i & 1))}, @ONES@, {v(P + i)}")
            return
        else:
            p(f"v_add_f32 {v(PS0)}, {v(PS0)}, {v(t0)}")
            p(f"v_add_f32 {v(PS1)}, {v(PS1)}, {v(t1)}")
        if not LATE_CVT or i < 4:
            p(f"@CVT@ {v(P + i)}, {v(t0)}, {v(t1)}")

    exps(0)
    for i in range(1, 16):
        exps(i)
        fin(i - 1)
        if hooks and i in hooks:
            hooks[i]()
    p("s_nop 0")
    fin(15)
    if SUM_MODE == "dot2":
        p("s_nop 2")           # dot result -> a different VALU opcode: 3 wait states
    p(f"v_add_f32 {v(PSUM)}, {v(PS0)}, {v(PS1)}")


def softmax_phase(p, tag, hooks=None):
    """fast path + slow path of one tile; ends with l += row sum.  hooks: instruction groups (the tile's DMA pieces) spliced into
    the fast path's exponential stream - every tile runs the fast path (tile 0 then always continues into the slow one)."""
    if hooks is None:                  # nothing else rides in the fast path: tile 0 goes straight to the slow one
        p(f"s_cmp_eq_u32 {OP['first']}, 1")
        p(f"s_cbranch_scc1 {Prog.L('slow' + tag)}")
    exp_sum_pack(p, hooks=hooks)
    if hooks is not None:
        p(f"s_cmp_eq_u32 {OP['first']}, 1")
        p(f"s_cbranch_scc1 {Prog.L('slow' + tag)}")
    p("s_nop 0")
    p(f"v_cmp_gt_f32_e32 vcc, {hex(SUM_MAX)}, {v(PSUM)}")
    p("s_cmp_eq_u64 vcc, exec")
    p(f"s_cbranch_scc1 {Prog.L('done' + tag)}")
    p.label("slow" + tag)
    # row maximum of the lane's 32 scores, then across the two half-waves that share a query
    p(f"v_max_f32 {v(MX)}, {v(S)}, {v(S + 1)}")
    for i in range(2, 32, 2):
        p(f"v_max3_f32 {v(MX)}, {v(MX)}, {v(S + i)}, {v(S + i + 1)}")
    p(f"v_mov_b32 {v(DLT)}, {v(MX)}")
    p("s_nop 1")
    p(f"v_permlane32_swap_b32 {v(DLT)}, {v(MX)}")
    p("s_nop 1")
    p(f"v_max_f32 {v(DLT)}, {v(DLT)}, {v(MX)}")
    p(f"s_cmp_eq_u32 {OP['first']}, 1")
    p(f"s_cbranch_scc1 {Prog.L('norescale' + tag)}")
    # the reference never goes down: delta = max(mx, 0); O and l move to the new reference
    p(f"v_max_f32 {v(DLT)}, {v(DLT)}, 0")
    p(f"v_exp_f32_e64 {v(ALPHA)}, -{v(DLT)}")
    p(f"v_add_f32 {v(MC)}, {v(MC)}, {v(DLT)}")
    p("s_nop 0")
    p(f"v_mul_f32 {v(LRUN)}, {v(LRUN)}, {v(ALPHA)}")
    for i in range(32):
        p(f"v_mul_f32 {v(O + i)}, {v(O + i)}, {v(ALPHA)}")
    p(f"s_branch {Prog.L('refdone' + tag)}")
    p.label("norescale" + tag)
    p(f"v_add_f32 {v(MC)}, {v(MC)}, {v(DLT)}")
    p.label("refdone" + tag)
    p(f"v_sub_f32 {v(NEGM)}, 0, {v(MC)}")
    for i in range(1, 16):
        p(f"v_mov_b32 {v(NEGM + i)}, {v(NEGM)}")
    exp_sum_pack(p, src_sub=DLT)
    p(f"s_mov_b32 {OP['first']}, 0")
    p.label("done" + tag)
    p(f"v_add_f32 {v(LRUN)}, {v(LRUN)}, {v(PSUM)}")


def stamp(p, k):
    """debug builds: acc_k += cycles since the previous stamp (drains the LDS / SMEM counter: a measuring build, not a fast one)"""
    if not DBG:
        return
    p("s_memtime vcc")
    p("s_waitcnt lgkmcnt(0)")
    p(f"s_sub_u32 {OP['tprev']}, vcc_lo, {OP['tprev']}")
    p(f"s_add_u32 {OP['acc%d' % k]}, {OP['acc%d' % k]}, {OP['tprev']}")
    p(f"s_mov_b32 {OP['tprev']}, vcc_lo")


DMA_IN = os.environ.get("ATTN_DMA", "m")        # where a wave issues its two DMA pieces per tile: "v0" = first thing in the softmax
                                                # phase, "vmid" = inside the exponential stream, "m" = behind MFMAs of the MFMA phase
DIST = 3 if DMA_IN == "m" else 2                # stages ahead (a buffer is free two phases after group B's MFMA phase read it)


def body(p, b):
    """tile t = 4k + b:  [softmax(t) + the DMA of stage t+2 | wait for this wave's older DMA pieces | fragment prefetch] barrier
    [P.V(t), K.Q^T(t+1)] barrier.  Stage t sits in buffer b; stage t+2 goes to buffer (b+2) % 4 (last read two phases ago).  Both
    groups run the same code, B one phase behind A.  `last` = this is the final trip: tiles t+DIST (b >= 4-DIST), t+DIST+1 and, for
    b == 3, t+1 do not exist then.  (DMA_IN == "m": stage t+3 from inside the MFMA phase.)"""
    tag = f"_{b}"
    L = Prog.L
    buf = (b + DIST) % NB
    vc = "nodv" + tag if b >= NB - DIST else None
    kc = "nodk" + tag if b >= NB - DIST - 1 else None

    def piece(which, lab):
        def f():
            if lab:
                p(f"s_cmp_eq_u32 {OP['last']}, 1")
                p(f"s_cbranch_scc1 {L(lab)}")
            dma_piece(p, buf, which)
            if lab:
                p.label(lab)
        return f

    hooks = None
    if DMA_IN == "v0":
        piece("v", vc)()
        piece("k", kc)()
    elif DMA_IN == "vmid":
        hooks = {5: piece("v", vc), 11: piece("k", kc)}
    softmax_phase(p, tag, hooks)
    stamp(p, 0)
    # this wave's pieces of the stage before must have landed; the two just issued may stay in flight
    p(f"s_cmp_eq_u32 {OP['last']}, 1")
    p(f"s_cbranch_scc1 {L('w0' + tag)}")
    p("s_waitcnt vmcnt(2)")
    p(f"s_branch {L('w1' + tag)}")
    p.label("w0" + tag)
    p("s_waitcnt vmcnt(0)")
    p.label("w1" + tag)
    if b == NB - 1:
        p(f"s_cmp_eq_u32 {OP['last']}, 1")
        p(f"s_cbranch_scc1 {L('final')}")
    prefetch(p, b, True)
    p("s_barrier")
    stamp(p, 1)
    p("s_setprio 1")
    mfma_phase(p, b, True, dma=(buf, [("v", vc), ("k", kc)]) if DMA_IN == "m" else None)
    p("s_setprio 0")
    stamp(p, 2)
    p("s_barrier")
    stamp(p, 3)


# One workgroup barrier per tile instead of two (body_onebar): 140.2 -> 134.3 us on the 4-head launch (1 023 TF/s), 198.0 -> 190.8
# on the 5-head one.  ATTN_ONEBAR=0 / the ATTN_DBG=1 timer build keep the two-barrier program.
ONEBAR = int(os.environ.get("ATTN_ONEBAR", "1")) and not DBG


def body_onebar(p, b, grp):
    """tile t = 4k + b with ONE barrier per tile.  Group A:  softmax(t) | barrier #t | MFMA(t)   (so between barriers it runs
    MFMA(t) then softmax(t+1)); group B:  barrier #t | softmax(t) | MFMA(t)   (softmax first, then MFMA): the two waves of a SIMD
    are complementary inside every interval and the softmax -> MFMA hand-over floats instead of waiting for the slower phase
    twice per tile.  LDS safety needs no second barrier with four stage buffers: stage t+3 (issued inside MFMA(t)) overwrites stage
    t-1, whose readers - MFMA(t-1) of both groups - finished before barrier #t; a wave waits for its own DMA pieces but the last
    two at the end of every softmax phase, which puts every stage's pieces in LDS at least one barrier before its first reader.
    (No fragment prefetch in front of a barrier here: group A would read a stage whose last pieces group B only waits for in
    the same interval.)"""
    tag = f"_{b}{'ab'[grp]}"
    L = Prog.L
    buf = (b + DIST) % NB
    vc = "nodv" + tag if b >= NB - DIST else None
    kc = "nodk" + tag if b >= NB - DIST - 1 else None
    if grp == 1:
        p("s_barrier")
    softmax_phase(p, tag)
    p(f"s_cmp_eq_u32 {OP['last']}, 1")
    p(f"s_cbranch_scc1 {L('w0' + tag)}")
    p("s_waitcnt vmcnt(2)")
    p(f"s_branch {L('w1' + tag)}")
    p.label("w0" + tag)
    p("s_waitcnt vmcnt(0)")
    p.label("w1" + tag)
    if grp == 0:
        p("s_barrier")
    if b == NB - 1:
        p(f"s_cmp_eq_u32 {OP['last']}, 1")
        p(f"s_cbranch_scc1 {L('final')}")
    p("s_setprio 1")
    mfma_phase(p, b, True, dma=(buf, [("v", vc), ("k", kc)]))
    p("s_setprio 0")


def program_onebar():
    global NPRE
    assert DMA_IN == "m"
    npre, NPRE = NPRE, 0
    p = Prog()
    L = Prog.L
    program_init(p)
    p("s_waitcnt vmcnt(0)")
    p("s_barrier")
    qk_only(p, NB - 1)
    p(f"s_cmp_eq_u32 {OP['grp']}, 0")
    p(f"s_cbranch_scc0 {L('loop_b')}")
    for grp in (0, 1):
        p.label("loop_" + "ab"[grp])
        p(f"s_add_u32 {OP['last']}, {OP['t']}, {NB}")
        p(f"s_cmp_ge_u32 {OP['last']}, {OP['nt']}")
        p(f"s_cselect_b32 {OP['last']}, 1, 0")
        for b in range(NB):
            body_onebar(p, b, grp)
        p(f"s_add_u32 {OP['t']}, {OP['t']}, {NB}")
        p(f"s_branch {L('loop_' + 'ab'[grp])}")
    p.label("final")
    p("s_setprio 1")
    mfma_phase(p, NB - 1, False)
    p("s_setprio 0")
    program_epilogue(p)
    NPRE = npre
    return p.lines


def program_init(p):
    # ---- state
    for i in range(32):
        p(f"v_mov_b32 {v(O + i)}, 0")
    for i in range(16):
        p(f"v_mov_b32 {v(NEGM + i)}, 0")
    p(f"v_mov_b32 {v(LRUN)}, 0")
    p(f"v_mov_b32 {v(MC)}, 0")
    for f in range(2):
        for s in range(4):
            for dst, src in ((KA, OP[f"ka{f}"]), (VA, OP[f"va{f}"])):
                p(f"v_xor_b32 {v(dst + 4 * f + s)}, {hex(s << 5)}, {src}" if s else f"v_mov_b32 {v(dst + 4 * f + s)}, {src}")
    p(f"s_mov_b32 {OP['t']}, 0")
    p(f"s_mov_b32 {OP['first']}, 1")
    p(f"s_mov_b32 {OP['vso']}, {hex(0x80 * DIST)}")       # next stage to issue is DIST: V^T(DIST), K(DIST + 1)
    p(f"s_mul_i32 {OP['kso']}, {OP['kstep']}, {DIST + 1}")
    if DBG:
        for k in range(4):
            p(f"s_mov_b32 {OP['acc%d' % k]}, 0")
        p("s_memtime vcc")
        p("s_waitcnt lgkmcnt(0)")
        p(f"s_mov_b32 {OP['tprev']}, vcc_lo")
        p(f"s_mov_b32 {OP['cyc0']}, vcc_lo")
        p("s_memrealtime vcc")
        p("s_waitcnt lgkmcnt(0)")
        p(f"s_mov_b32 {OP['rt0']}, vcc_lo")


def program():
    if ONEBAR:
        return program_onebar()
    p = Prog()
    L = Prog.L
    program_init(p)
    # ---- stages -1 (K(0), buffer 3), 0 .. DIST-1 were issued by the caller
    p("s_waitcnt vmcnt(0)")
    p("s_barrier")
    p(f"s_cmp_eq_u32 {OP['grp']}, 0")
    p(f"s_cbranch_scc0 {L('pro_b')}")
    qk_only(p, NB - 1)
    p("s_barrier")
    p(f"s_branch {L('loop')}")
    p.label("pro_b")
    p("s_barrier")
    qk_only(p, NB - 1)
    p("s_barrier")
    # ---- tile loop, four tiles per trip
    p.label("loop")
    p(f"s_add_u32 {OP['last']}, {OP['t']}, {NB}")
    p(f"s_cmp_ge_u32 {OP['last']}, {OP['nt']}")
    p(f"s_cselect_b32 {OP['last']}, 1, 0")
    for b in range(NB):
        body(p, b)
    p(f"s_add_u32 {OP['t']}, {OP['t']}, {NB}")
    p(f"s_branch {L('loop')}")
    # ---- final tile: P.V only; A's last barrier lets B into this phase, B needs none behind it
    p.label("final")
    prefetch(p, NB - 1, False)
    p("s_barrier")
    p("s_setprio 1")
    mfma_phase(p, NB - 1, False)
    p("s_setprio 0")
    p(f"s_cmp_eq_u32 {OP['grp']}, 0")
    p(f"s_cbranch_scc0 {L('epi')}")
    p("s_barrier")
    p.label("epi")
    program_epilogue(p)
    return p.lines


def program_epilogue(p):
    L = Prog.L
    # ---- epilogue: l = l(lower half-wave) + l(upper), O / l -> 16 bit, lane owns query row, channels f*32 + 8g + 4hf + e
    p("s_nop 7")
    p(f"v_mov_b32 {v(PSUM)}, {v(LRUN)}")
    p("s_nop 1")
    p(f"v_permlane32_swap_b32 {v(PSUM)}, {v(LRUN)}")
    p("s_nop 1")
    p(f"v_add_f32 {v(LRUN)}, {v(LRUN)}, {v(PSUM)}")
    # partial mode (this workgroup saw half of the keys): the un-normalised fp32 row + (m, l) go to the workspace, rows of 68 floats
    p(f"s_cmp_eq_u32 {OP['part']}, 1")
    p(f"s_cbranch_scc0 {L('epi_full')}")
    for f in range(2):
        for g in range(4):
            p(f"global_store_dwordx4 {OP['optr']}, {v(O + 16 * f + 4 * g, 4)}, off offset:{(f * 32 + 8 * g) * 4}")
    p(f"v_mov_b32 {v(TMP)}, {v(MC)}")
    p(f"v_mov_b32 {v(TMP + 1)}, {v(LRUN)}")
    p("s_mov_b64 exec, 0xffffffff")                    # the lower half-wave's pointer is the row start
    p("s_nop 1")
    p(f"global_store_dwordx2 {OP['optr']}, {v(TMP, 2)}, off offset:256")
    p("s_mov_b64 exec, -1")
    p(f"s_branch {L('end')}")
    p.label("epi_full")
    p("s_nop 0")
    p(f"v_rcp_f32 {v(ALPHA)}, {v(LRUN)}")
    p("s_nop 0")
    for f in range(2):
        for g in range(4):
            for e in range(4):
                p(f"v_mul_f32 {v(O + 16 * f + 4 * g + e)}, {v(O + 16 * f + 4 * g + e)}, {v(ALPHA)}")
            d = S + 2 * (4 * f + g)
            p(f"@CVT@ {v(d)}, {v(O + 16 * f + 4 * g)}, {v(O + 16 * f + 4 * g + 1)}")
            p(f"@CVT@ {v(d + 1)}, {v(O + 16 * f + 4 * g + 2)}, {v(O + 16 * f + 4 * g + 3)}")
    for f in range(2):
        for g in range(4):
            d = S + 2 * (4 * f + g)
            p(f"global_store_dwordx2 {OP['optr']}, {v(d, 2)}, off offset:{(f * 32 + 8 * g) * 2}" + (" sc1" if OUT_WT else ""))
    p.label("end")
    if DBG:          # lanes 0..31 overwrite the first 24 bytes of their output row with the wave's timers
        p("s_waitcnt vmcnt(0)")
        p("s_memtime vcc")
        p("s_waitcnt lgkmcnt(0)")
        p(f"s_sub_u32 {OP['cyc0']}, vcc_lo, {OP['cyc0']}")
        p("s_memrealtime vcc")
        p("s_waitcnt lgkmcnt(0)")
        p(f"s_sub_u32 {OP['rt0']}, vcc_lo, {OP['rt0']}")
        p("s_mov_b64 exec, 0xffffffff")
        p(f"v_mov_b32 {v(TMP + 4)}, {OP['cyc0']}")
        p(f"v_mov_b32 {v(TMP + 5)}, {OP['rt0']}")
        p("s_nop 1")
        p(f"global_store_dwordx2 {OP['optr']}, {v(TMP + 4, 2)}, off offset:16")
        for k in range(4):
            p(f"v_mov_b32 {v(TMP + k)}, {OP['acc%d' % k]}")
        p("s_nop 1")
        p(f"global_store_dwordx4 {OP['optr']}, {v(TMP, 4)}, off")
        p("s_waitcnt vmcnt(0)")
        p("s_mov_b64 exec, -1")
    return p.lines


ABL = int(os.environ.get("ATTN_ABL", "0"))     # timing-only ablations (tools/ab_attn.sh): 1 = v_mov for v_exp, 2 = no MFMAs, 3 = no fragment
                                               # reads, 4 = no softmax arithmetic at all, 5 = no s_setprio, 6 = no DMA in the loop


def ablate(lines):
    out = []
    for ln in lines:
        if ABL in (7, 8) and ln == "s_barrier":          # 7 = no softmax arithmetic and no barriers, 8 = everything but the barriers
            continue
        if ABL == 7 and ln.startswith(f"v_add_f32 {v(PSUM)}, {v(PS0)}"):
            ln = f"v_mov_b32 {v(PSUM)}, 0"
        if ABL == 7 and (ln.startswith("@EXP@") or ln.startswith("@CVT@") or ln.startswith("v_add_f32") or ln.startswith("v_sub_f32")):
            continue
        if ABL in (2, 3, 4, 7, 8):      # garbage scores must not reach the slow path: always take the fast one
            if ln == "s_cmp_eq_u64 vcc, exec":
                ln = "s_cmp_eq_u32 0, 0"
            if ln == f"s_mov_b32 {OP['first']}, 1":
                ln = f"s_mov_b32 {OP['first']}, 0"
        if ABL == 1 and ln.startswith("@EXP@"):
            ln = "v_mov_b32" + ln[5:]
        if ABL == 2 and ln.startswith("@MN@"):
            continue
        if ABL == 3 and (ln.startswith("ds_read") or ln.startswith("s_waitcnt lgkmcnt")):
            continue
        if ABL == 4 and ln.startswith(f"v_add_f32 {v(PSUM)}, {v(PS0)}"):
            ln = f"v_mov_b32 {v(PSUM)}, 0"
        if ABL == 4 and (ln.startswith("@EXP@") or ln.startswith("@CVT@") or ln.startswith("v_add_f32") or ln.startswith("v_sub_f32")):
            continue
        if ABL == 5 and ln.startswith("s_setprio"):
            continue
        if ABL == 6 and ln.startswith("buffer_load"):
            continue
        out.append(ln)
    return out


def emit(lines, name):
    out = [f"#define {name}(MN, CVT, DOT, ONES) \\"]
    body_lines = []
    lines = ablate(lines) if ABL else lines
    for ln in lines:
        if ln.startswith("@MN@"):
            body_lines.append('  MN "' + ln[4:] + '\\n\\t"')
        elif ln.startswith("@CVT@"):
            body_lines.append('  CVT "' + ln[5:] + '\\n\\t"')
        elif ln.startswith("@DOT@"):
            a, b = ln[5:].split("@ONES@")
            body_lines.append('  DOT "' + a + '" ONES "' + b + '\\n\\t"')
        elif ln.startswith("@EXP@"):
            body_lines.append('  "v_exp_f32' + ln[5:] + '\\n\\t"')
        else:
            body_lines.append('  "' + ln + '\\n\\t"')
    return out[0] + "\n" + " \\\n".join(body_lines) + "\n"


def main():
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "unirestore_amd", "csrc", "attention_pp_asm.inc")
    txt = "// GENERATED by tools/gen_attn_asm.py - do not edit (tile loop + epilogue of attn_pp64_kernel; see the generator's docstring)\n#pragma once\n\n"
    txt += "// operands: " + ", ".join(f"%{i} {n}" for i, n in enumerate(OPERANDS)) + "\n"
    txt += f"// hand-owned registers v{LO}..v255: S v{S}, O v{O}, NEGM v{NEGM}, RING v{RING}, P v{P}, KA v{KA}, VA v{VA}, TMP v{TMP}\n"
    txt += f"#define ATTN_PP_DIST {DIST}      // stages the caller issues ahead: -1 .. DIST-1\n"
    txt += emit(program(), "ATTN_PP_ASM")
    txt += "\n#define ATTN_PP_CLOBBERS " + ", ".join(f'"v{i}"' for i in range(LO, 256)) + ', "vcc", "scc", "m0", "memory"\n'
    open(out, "w").write(txt)
    print("wrote", os.path.normpath(out), f"({len(program())} instructions, hand-owned v{LO}..v255)")


if __name__ == "__main__":
    main()
