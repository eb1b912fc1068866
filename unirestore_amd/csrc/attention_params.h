// Launch parameters shared by the attention translation units (attention.hip: d = 64 / 128, attention512.hip: d = 512).
#pragma once
#include <cstdint>

struct AttnP {
  const uint16_t* q; const uint16_t* k; const uint16_t* vt; uint16_t* o;
  int B, H, Tq, Tk, ldq, ldk, ldvt, ldo;
  long long bs_q, bs_k, bs_vt, bs_o;
  float scale_log2e;
  // ping-pong kernel (attention_pp.hip) only: workgroups [0, n_full) take a whole (batch-head, 256-query tile) each; the rest come in
  // pairs that split the keys of one tile in halves and leave un-normalised fp32 rows + (m, l) in ws for attn_pp_combine_kernel
  int n_full;
  float* ws;
};

// ONE rule for the ping-pong kernel's grid (used by ur_attention_workspace_bytes, the dispatcher's fill estimate and the launch):
// n = (Tq / 256) * B * H query tiles; when n = whole rounds of the 256 CUs + r with 0 < r <= 128, the last r tiles are split in two
// key halves (2r <= 256 workgroups fill the last round) - if a workspace of attn_pp_ws_bytes(r) is there and Tk % 512 == 0.
// Returns r (0: unsplit).  UR_ATTN_NOSPLIT=1 disables the split everywhere.
#include <cstdlib>
static inline size_t attn_pp_ws_bytes(long long r) { return (size_t)r * 2 * 256 * 68 * 4; }
static inline long long attn_pp_split_tiles(int B, int H, int Tq, int Tk, int D) {
  static const bool nosplit = getenv("UR_ATTN_NOSPLIT") && atoi(getenv("UR_ATTN_NOSPLIT")) != 0;
  if (nosplit || D != 64 || Tq % 256 || Tk % 512) return 0;
  const long long n = (long long)(Tq / 256) * B * H, full = n / 256 * 256, r = n - full;
  return (full >= 256 && r > 0 && r <= 128) ? r : 0;
}
// What the ping-pong kernel's hand-written loop assumes (it has no tails and its buffer descriptors do no range check):
// whole 256-query / 256-key tiles and 16-byte aligned rows.
static inline bool attn_pp_shape_ok(const AttnP& p, int D) {
  return D == 64 && p.Tq % 256 == 0 && p.Tk % 256 == 0 && p.Tk >= 256 && p.ldq % 8 == 0 && p.ldk % 8 == 0 && p.ldvt % 8 == 0 && p.ldo % 4 == 0 &&
         p.bs_q % 8 == 0 && p.bs_k % 8 == 0 && p.bs_vt % 8 == 0 &&
         (((unsigned long long)p.q | (unsigned long long)p.k | (unsigned long long)p.vt) & 15ull) == 0 && ((unsigned long long)p.o & 7ull) == 0;
}
