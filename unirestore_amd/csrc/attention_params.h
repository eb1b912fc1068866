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
