// Launch parameters shared by the attention translation units (attention.hip: d = 64 / 128, attention512.hip: d = 512).
#pragma once
#include <cstdint>

struct AttnP {
  const uint16_t* q; const uint16_t* k; const uint16_t* vt; uint16_t* o;
  int B, H, Tq, Tk, ldq, ldk, ldvt, ldo;
  long long bs_q, bs_k, bs_vt, bs_o;
  float scale_log2e;
};
