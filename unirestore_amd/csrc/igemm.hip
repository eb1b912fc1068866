// Implicit-GEMM convolution / linear for gfx950: NHWC bf16 activations, [Cout][KH*KW*Cin] bf16 weights,
// v_mfma_f32_32x32x16_bf16 with fp32 accumulate, fused epilogues.  See include/unirestore_hip.h.
//
// Mapping (MI355X-first, not a cuDNN-style port):
//   * GEMM view: M = N*OH*OW output pixels, N = Cout, K = KH*KW*Cin; the K index runs (tap, cin) so every
//     16-byte vector a lane loads is 8 contiguous input channels of ONE pixel (NHWC) or of one weight row.
//   * MFMA "A" operand = weight rows (cout), "B" operand = pixels, so each lane of the 32x32 accumulator
//     holds 4 CONSECUTIVE output channels of one pixel -> 8-byte bf16 stores straight into NHWC.
//   * 256 threads = 4 waves; register-staged global->LDS pipeline (issue tile t+1 loads, compute tile t,
//     write LDS, one barrier per 64-deep K tile), 2 LDS stages.
//   * LDS tile = [row][128 B] with the 16-B slot XOR-swizzled by (row>>1)&7: conflict-free for both the
//     ds_write_b128 staging pattern and the ds_read_b128 fragment pattern of 32-row MFMA operands.
//   * zero padding, stride, nearest-2x upsample and channel concat are address arithmetic in the loader.
//   * block id -> tile map is XCD-aware (blocks that share an activation tile land on one XCD's L2).
#include "igemm_impl.h"

namespace urk {   // launcher instantiations live in igemm_v1a/v1b/v2/halo/g1.hip, one object per 16-bit type (parallel compilation)
int v1_128x128_bf16(void* kp, hipStream_t s);
int v1_128x128_f16(void* kp, hipStream_t s);
static inline int v1_128x128(void* kp, hipStream_t s) { return static_cast<ConvK*>(kp)->f16 ? v1_128x128_f16(kp, s) : v1_128x128_bf16(kp, s); }
int v1_128x160_bf16(void* kp, hipStream_t s);
int v1_128x160_f16(void* kp, hipStream_t s);
static inline int v1_128x160(void* kp, hipStream_t s) { return static_cast<ConvK*>(kp)->f16 ? v1_128x160_f16(kp, s) : v1_128x160_bf16(kp, s); }
int v1_128x64_bf16(void* kp, hipStream_t s);
int v1_128x64_f16(void* kp, hipStream_t s);
static inline int v1_128x64(void* kp, hipStream_t s) { return static_cast<ConvK*>(kp)->f16 ? v1_128x64_f16(kp, s) : v1_128x64_bf16(kp, s); }
int v1_256x32_bf16(void* kp, hipStream_t s);
int v1_256x32_f16(void* kp, hipStream_t s);
static inline int v1_256x32(void* kp, hipStream_t s) { return static_cast<ConvK*>(kp)->f16 ? v1_256x32_f16(kp, s) : v1_256x32_bf16(kp, s); }
int v1_64x64_bf16(void* kp, hipStream_t s);
int v1_64x64_f16(void* kp, hipStream_t s);
static inline int v1_64x64(void* kp, hipStream_t s) { return static_cast<ConvK*>(kp)->f16 ? v1_64x64_f16(kp, s) : v1_64x64_bf16(kp, s); }
int v2_256x32_bf16(void* kp, hipStream_t s);
int v2_256x32_f16(void* kp, hipStream_t s);
static inline int v2_256x32(void* kp, hipStream_t s) { return static_cast<ConvK*>(kp)->f16 ? v2_256x32_f16(kp, s) : v2_256x32_bf16(kp, s); }
int v2_128x64_bf16(void* kp, hipStream_t s);
int v2_128x64_f16(void* kp, hipStream_t s);
static inline int v2_128x64(void* kp, hipStream_t s) { return static_cast<ConvK*>(kp)->f16 ? v2_128x64_f16(kp, s) : v2_128x64_bf16(kp, s); }
int v2_256x160_bf16(void* kp, hipStream_t s);
int v2_256x160_f16(void* kp, hipStream_t s);
static inline int v2_256x160(void* kp, hipStream_t s) { return static_cast<ConvK*>(kp)->f16 ? v2_256x160_f16(kp, s) : v2_256x160_bf16(kp, s); }
int v2_256x128_bf16(void* kp, hipStream_t s);
int v2_256x128_f16(void* kp, hipStream_t s);
static inline int v2_256x128(void* kp, hipStream_t s) { return static_cast<ConvK*>(kp)->f16 ? v2_256x128_f16(kp, s) : v2_256x128_bf16(kp, s); }
int gemm_256x256_bf16(void* kp, hipStream_t s);
int gemm_256x256_f16(void* kp, hipStream_t s);
static inline int gemm_256x256(void* kp, hipStream_t s) { return static_cast<ConvK*>(kp)->f16 ? gemm_256x256_f16(kp, s) : gemm_256x256_bf16(kp, s); }
int gemm_256x320_pair_bf16(void* kp, hipStream_t s);
int gemm_256x320_pair_f16(void* kp, hipStream_t s);
static inline int gemm_256x320_pair(void* kp, hipStream_t s) { return static_cast<ConvK*>(kp)->f16 ? gemm_256x320_pair_f16(kp, s) : gemm_256x320_pair_bf16(kp, s); }
int g1_128x128_bf16(void* kp, hipStream_t s);
int g1_128x128_f16(void* kp, hipStream_t s);
static inline int g1_128x128(void* kp, hipStream_t s) { return static_cast<ConvK*>(kp)->f16 ? g1_128x128_f16(kp, s) : g1_128x128_bf16(kp, s); }
int g1_128x160_bf16(void* kp, hipStream_t s);
int g1_128x160_f16(void* kp, hipStream_t s);
static inline int g1_128x160(void* kp, hipStream_t s) { return static_cast<ConvK*>(kp)->f16 ? g1_128x160_f16(kp, s) : g1_128x160_bf16(kp, s); }
int g1_128x64_bf16(void* kp, hipStream_t s);
int g1_128x64_f16(void* kp, hipStream_t s);
static inline int g1_128x64(void* kp, hipStream_t s) { return static_cast<ConvK*>(kp)->f16 ? g1_128x64_f16(kp, s) : g1_128x64_bf16(kp, s); }
int g1_64x64_bf16(void* kp, hipStream_t s);
int g1_64x64_f16(void* kp, hipStream_t s);
static inline int g1_64x64(void* kp, hipStream_t s) { return static_cast<ConvK*>(kp)->f16 ? g1_64x64_f16(kp, s) : g1_64x64_bf16(kp, s); }
int g1_64x64_deep_bf16(void* kp, hipStream_t s);
int g1_64x64_deep_f16(void* kp, hipStream_t s);
static inline int g1_64x64_deep(void* kp, hipStream_t s) { return static_cast<ConvK*>(kp)->f16 ? g1_64x64_deep_f16(kp, s) : g1_64x64_deep_bf16(kp, s); }
int g1_128x64_deep_bf16(void* kp, hipStream_t s);
int g1_128x64_deep_f16(void* kp, hipStream_t s);
static inline int g1_128x64_deep(void* kp, hipStream_t s) { return static_cast<ConvK*>(kp)->f16 ? g1_128x64_deep_f16(kp, s) : g1_128x64_deep_bf16(kp, s); }
int halo_8x32_160_bf16(void* kp, hipStream_t s);
int halo_8x32_160_f16(void* kp, hipStream_t s);
static inline int halo_8x32_160(void* kp, hipStream_t s) { return static_cast<ConvK*>(kp)->f16 ? halo_8x32_160_f16(kp, s) : halo_8x32_160_bf16(kp, s); }
int halo_8x32_128_bf16(void* kp, hipStream_t s);
int halo_8x32_128_f16(void* kp, hipStream_t s);
static inline int halo_8x32_128(void* kp, hipStream_t s) { return static_cast<ConvK*>(kp)->f16 ? halo_8x32_128_f16(kp, s) : halo_8x32_128_bf16(kp, s); }
int halo_thin_32_bf16(void* kp, hipStream_t s);
int halo_thin_32_f16(void* kp, hipStream_t s);
static inline int halo_thin_32(void* kp, hipStream_t s) { return static_cast<ConvK*>(kp)->f16 ? halo_thin_32_f16(kp, s) : halo_thin_32_bf16(kp, s); }
int himg_16x16_bf16(void* kp, hipStream_t s);
int himg_16x16_f16(void* kp, hipStream_t s);
static inline int himg_16x16(void* kp, hipStream_t s) { return static_cast<ConvK*>(kp)->f16 ? himg_16x16_f16(kp, s) : himg_16x16_bf16(kp, s); }
int himg_8x8x4_bf16(void* kp, hipStream_t s);
int himg_8x8x4_f16(void* kp, hipStream_t s);
static inline int himg_8x8x4(void* kp, hipStream_t s) { return static_cast<ConvK*>(kp)->f16 ? himg_8x8x4_f16(kp, s) : himg_8x8x4_bf16(kp, s); }
int wstream_8x8_bf16(void* kp, hipStream_t s);
int wstream_8x8_f16(void* kp, hipStream_t s);
static inline int wstream_8x8(void* kp, hipStream_t s) { return static_cast<ConvK*>(kp)->f16 ? wstream_8x8_f16(kp, s) : wstream_8x8_bf16(kp, s); }
#ifdef UR_AB_VARIANTS
int g1_ab_bf16(void* kp, hipStream_t s, int id);
#endif
}  // namespace urk

namespace {

int dispatch_conv(ConvK& k, hipStream_t s, bool pair) {
  k.patch_tw = 0;
  static const bool no_halo = getenv("UR_IGEMM_NOHALO") != nullptr;
  if (!no_halo && k.KH == 3 && k.stride == 1 && k.pad_t == 1 && k.pad_l == 1 && k.kcm && k.staged_ok_ && !pair && k.nbatch == 1 &&
      k.OW % 32 == 0 && k.OH % 8 == 0 && k.OH == (k.ups ? 2 * k.H : k.H) && k.OW == (k.ups ? 2 * k.W : k.W) && !k.yt) {
    // tile width: 128 or 160 output channels, whichever divides Cout; when both do, the one whose tile count fills whole
    // rounds of 256 CUs better (e.g. 1280 channels at 32 x 32: 8 x 160 -> 256 tiles = one round, 10 x 128 -> 320 = two)
    const bool ok128 = k.Cout % 128 == 0, ok160 = k.Cout % 160 == 0;
    const long long tm8 = (long long)k.N * (k.OH / 8) * (k.OW / 32);
    // (<= 128 tiles are split over channel chunks into floor(256 / tiles) workgroups each)
    auto eff = [](long long t) { return t <= 128 ? (double)(t * (256 / t)) / 256.0 : (double)t / (double)(((t + 255) / 256) * 256); };
    static const bool old_w = getenv("UR_IGEMM_OLDW") != nullptr;
    bool use160 = ok160 && (!ok128 || (!old_w && eff(tm8 * (k.Cout / 160)) >= eff(tm8 * (k.Cout / 128))));
    const long long tiles8 = tm8 * (use160 ? k.Cout / 160 : k.Cout / 128);
    if ((ok128 || ok160) && tiles8 >= 64) {
      if (use160) return urk::halo_8x32_160(&k, s);
      return urk::halo_8x32_128(&k, s);
    }
  }
  // conv_out layers: <= 32 output channels (fp32 or 16-bit) of a 3x3 / stride 1 / pad 1 conv with chunk-major weights
  static const bool no_thin = getenv("UR_IGEMM_NOTHIN") != nullptr;
  if (!no_thin && !no_halo && k.KH == 3 && k.stride == 1 && k.pad_t == 1 && k.pad_l == 1 && k.kcm && !pair && k.nbatch == 1 && !k.ups && k.C2 == 0 &&
      k.Cout <= 32 && k.OW % 32 == 0 && k.OH % 8 == 0 && k.OH == k.H && k.OW == k.W && !k.yt && !k.gn_part && !k.gn_ab && !k.row_stats && !k.ln_stats &&
      !k.bias_img && (long long)k.N * (k.OH / 8) * (k.OW / 32) >= 128)
    return urk::halo_thin_32(&k, s);
  static const bool no_himg = getenv("UR_IGEMM_NOHIMG") != nullptr;
  static const bool no_himg_ups = getenv("UR_IGEMM_NOHIMGUPS") != nullptr;
  // (round 6: the 8 x 8 -> 16 x 16 upsampling conv runs on the whole-image tile too - the patch pieces read their nearest source pixel)
  const bool himg_ups = !no_himg_ups && k.ups && k.OH == 16 && k.OW == 16 && k.H == 8 && k.W == 8;
  if (!no_himg && k.KH == 3 && k.stride == 1 && k.pad_t == 1 && k.pad_l == 1 && k.kcm && k.staged_ok_ && !pair && k.nbatch == 1 &&
      ((!k.ups && k.OH == k.H && k.OW == k.W) || himg_ups) && !k.yt && k.Cout % 128 == 0 && k.nk >= 36) {
    // 8 x 8 maps of a few images: a weight stream (csrc/conv_wstream.hip) when the caller packed the fragment-major copy
    static const bool no_wstream = getenv("UR_IGEMM_NOWSTREAM") != nullptr;
    if (!no_wstream && k.wf && k.OH == 8 && k.OW == 8 && k.N <= 16 && k.Cin % 256 == 0 && k.nk >= 72 && k.y && k.ws && !k.gn_ab && !k.row_stats && !k.ln_stats &&
        (long long)(k.nk / 36) * k.M * k.Cout * 4 <= (long long)k.ws_bytes_)
      return urk::wstream_8x8(&k, s);
    if (k.OH == 16 && k.OW == 16) return urk::himg_16x16(&k, s);
    if (k.OH == 8 && k.OW == 8 && k.N % 4 == 0) {
      return urk::himg_8x8x4(&k, s);
    }
  }
  static const bool no_g256 = getenv("UR_IGEMM_NOG256") != nullptr;
  if (!no_g256 && k.KH == 1 && k.stride == 1 && !k.ups && k.C2 == 0 && k.staged_ok_ && k.Cout % 256 == 0 && !k.yt && (k.act == UR_ACT_GEGLU || k.act == UR_ACT_GATE) &&
      (long long)((k.M + 255) / 256) * (k.Cout / 256) * k.nbatch >= 200 && (long long)k.M * k.ldx < (1ll << 31) - 256)
  {
    // 320-wide tiles when they land on whole rounds of CUs (2048 x 10240: 8 x 32 = 256 tiles instead of 320)
    static const bool no_g320 = getenv("UR_IGEMM_NOG320") != nullptr;
    auto fill = [](long long t) { return (double)t / (double)(((t + 255) / 256) * 256); };
    const long long tm = (k.M + 255) / 256;
    if (!no_g320 && k.Cout % 320 == 0 && k.nbatch == 1 && !k.bias_img && fill(tm * (k.Cout / 320)) > fill(tm * (k.Cout / 256)))
      return urk::gemm_256x320_pair(&k, s);
    return urk::gemm_256x256(&k, s);
  }
  static const bool force_v1 = getenv("UR_IGEMM_V1") != nullptr;
  // short-K GEMMs (<= 10 K tiles: per-workgroup prologue/epilogue latency dominates): 128-row tiles, 2 workgroups per CU
  const bool use_v1 = force_v1 || (k.KH == 1 && k.nk <= 10);
  const long long blocks128 = (long long)((k.M + 127) / 128) * ((k.Cout + 127) / 128) * k.nbatch;
  static const bool no_t64 = getenv("UR_IGEMM_NOT64") != nullptr;
  // pure GEMMs that will not be split: the LDS-DMA twin of the register-staged kernel (no staging registers / ds_writes)
  static const bool g1dma = getenv("UR_IGEMM_NOG1DMA") == nullptr;
  const bool g1 = g1dma && k.KH == 1 && k.stride == 1 && !k.ups && k.C2 == 0 && k.pad_t == 0 && k.pad_l == 0 && k.OH == k.H && k.OW == k.W &&
                  (long long)k.M * k.ldx + k.Ktot < (1ll << 31) - 256;
#ifdef UR_AB_VARIANTS
  if (const char* ab = getenv("UR_AB_ID")) if (g1 && !pair && !k.f16 && atoi(ab) >= 0) return urk::g1_ab_bf16(&k, s, atoi(ab));
#endif
  auto nosplit = [&](int bm, int bn) { return (long long)((k.M + bm - 1) / bm) * ((k.Cout + bn - 1) / bn) * k.nbatch >= 200 || k.nk < 8 || !k.ws; };
  if (!no_t64 && k.KH == 1 && !pair && k.Cout > 64) {
    // too few 128 x 128 tiles to fill 256 CUs and K too short for split-K to pay for its reduce pass: 64 x 64 tiles
    // (>= 128 such tiles run faster unsplit up to K = 3072 than split with a reduce pass: 512 x 1280 x 1280 11.3 vs 14.6 us,
    //  2048 x 1280 x 2560 29 vs 34 us - tools/ab_gemm_sweep.py)
    const long long blocks64 = (long long)((k.M + 63) / 64) * ((k.Cout + 63) / 64) * k.nbatch;
    // (grids of <= 256 workgroups - the 8x8 level - are latency-bound per K tile: four ring stages, and K up to 2560 stays unsplit:
    //  512 x 1280 x 1280 12.9 -> 9.1 us, x 2560 20.8 (split + reduce) -> 14.3 us; profiles/r5_wreg_ab.txt)
    static const bool no_deep = getenv("UR_IGEMM_NODEEP") != nullptr;
    if (!no_deep && blocks128 < 200 && g1 && blocks64 >= 128 && blocks64 <= 256 && k.nk >= 8 && k.nk <= 48) return urk::g1_64x64_deep(&k, s);
    // long-K GEMMs of the 16x16 level: 128 x 64 tiles, three stages, unsplit (2048 x 1280 x 2560 25.9 -> 23.3 us, x 5120 49.1 -> 43.7 us)
    if (!no_deep && g1 && blocks128 < 200 && k.nk > 24 && k.nk <= 96 && (long long)((k.M + 127) / 128) * ((k.Cout + 63) / 64) * k.nbatch >= 256)
      return urk::g1_128x64_deep(&k, s);
    if (blocks128 < 200 && g1 && ((blocks64 >= 128 && k.nk <= 24) || (blocks64 >= 256 && k.nk <= 48))) return urk::g1_64x64(&k, s);
    if (blocks128 < 200 && k.nk <= 24) return g1 && nosplit(64, 64) ? urk::g1_64x64(&k, s) : urk::v1_64x64(&k, s);
    // 1 < tiles/CU < 2 at 128 x 128: halve the N tile so every CU gets the same work
    if (use_v1 && blocks128 > 256 && blocks128 < 400 && k.Cout % 128 == 0) return g1 && nosplit(128, 64) ? urk::g1_128x64(&k, s) : urk::v1_128x64(&k, s);
  }
  if (use_v1) {
    if (pair) return urk::v1_128x128(&k, s);  // a|g 32-row blocks must sit in one wave tile
    if (k.Cout <= 32) return urk::v1_256x32(&k, s);
    if (k.Cout <= 64) return urk::v1_128x64(&k, s);
    if (k.Cout % 160 == 0 && k.Cout % 128 != 0) return g1 && nosplit(128, 160) ? urk::g1_128x160(&k, s) : urk::v1_128x160(&k, s);
    return g1 && nosplit(128, 128) ? urk::g1_128x128(&k, s) : urk::v1_128x128(&k, s);
  }
  // v2 (LDS-DMA ring).  One workgroup per CU: pick the 256-row / 8-wave tiles when they still fill the chip.
  if (k.Cout <= 32) return urk::v2_256x32(&k, s);
  if (k.Cout <= 64 && !pair) return urk::v2_128x64(&k, s);
  const bool n160 = !pair && k.Cout % 160 == 0 && k.Cout % 128 != 0;
  const long long big_tiles = (long long)((k.M + 255) / 256) * ((k.Cout + (n160 ? 159 : 127)) / (n160 ? 160 : 128)) * k.nbatch;
  static const bool no_fill = getenv("UR_IGEMM_NOFILL") != nullptr;
  if (!no_fill && g1 && !pair && big_tiles < 256) {
    // 256-row tiles would leave CUs without a workgroup (e.g. 8192 x 640: 160 tiles): 128-row LDS-DMA tiles, two per CU,
    // in the width that lands closest to whole CUs (8192 x 640 -> 64 x 4 tiles of 160 = 256)
    const long long t160 = k.Cout % 160 == 0 ? (long long)((k.M + 127) / 128) * (k.Cout / 160) : 0;
    const long long t128 = (long long)((k.M + 127) / 128) * ((k.Cout + 127) / 128);
    auto fill = [](long long t) { return t <= 0 ? 0.0 : (double)t / (double)(((t + 255) / 256) * 256); };
    if (t160 >= 200 && fill(t160) >= fill(t128)) return urk::g1_128x160(&k, s);
    if (t128 >= 200) return urk::g1_128x128(&k, s);
  }
  if (big_tiles >= 160) {
    // pure GEMMs onto 320/960-wide outputs: two 128 x 160 workgroups per CU beat one 256 x 160 (32768 x 320 x 1280: 41 vs 44.5 us)
    if (n160 && g1 && !no_fill) return urk::g1_128x160(&k, s);
    if (n160) return urk::v2_256x160(&k, s);
    return urk::v2_256x128(&k, s);
  }
  // small problems: the register-staged kernel (two workgroups per CU), split-K when few tiles meet a long K
  if (k.Cout % 160 == 0 && k.Cout % 128 != 0 && !pair) return urk::v1_128x160(&k, s);
  return urk::v1_128x128(&k, s);
}

}  // namespace

static int conv_impl(const ur_conv_desc* d, ur_stream_t stream, int dry, ur_conv_plan* plan) {
  UR_REQUIRE(d && d->x && d->w, "null x/w");
  UR_REQUIRE(d->KH == d->KW && (d->KH == 1 || d->KH == 3), "only 1x1 and 3x3 kernels");
  UR_REQUIRE(d->C1 > 0 && d->C1 % 8 == 0 && d->C2 % 8 == 0 && d->ldx % 8 == 0, "Cin/ldx must be multiples of 8");
  UR_REQUIRE(d->C2 == 0 || (d->x2 && d->ldx2 % 8 == 0), "virtual concat needs x2");
  UR_REQUIRE(d->Cout > 0 && d->Cout % 4 == 0 && d->ldw % 8 == 0, "Cout%4, ldw%8");
  UR_REQUIRE(d->nbatch >= 1 && d->stride >= 1 && d->OH > 0 && d->OW > 0, "bad dims");
  UR_REQUIRE(d->y || d->yt || d->gn_part, "no output requested");
  UR_REQUIRE_DT(d->dtype);
  const bool pair = d->act == UR_ACT_GEGLU || d->act == UR_ACT_GATE;
  if (pair && d->Cout % 64 != 0)     // (production widths are 128 / 256 / 512 output channels; refused, never computed wrongly)
    return ur::fail(UR_E_UNSUPPORTED, "ur_conv2d_nhwc: pair activations (GEGLU / SimpleGate) need Cout % 64 == 0 (32-row a|g interleave)");
  UR_REQUIRE(!d->y || (d->out_f32 ? d->ldy % 4 == 0 : d->ldy % 4 == 0), "ldy%4");
  UR_REQUIRE(!d->residual || d->ldr % 4 == 0, "ldr%4");
  UR_REQUIRE(!d->yt || (d->t_rows > 0 && d->n_split % 4 == 0 && !pair), "bad transposed-output spec");

  ConvK k;
  k.x = (const uint16_t*)d->x; k.x2 = (const uint16_t*)d->x2; k.w = (const uint16_t*)d->w; k.bias = d->bias;
  k.res = (const uint16_t*)d->residual; k.y = d->y; k.yt = (uint16_t*)d->yt;
  k.ws = d->workspace; k.ws_bytes_ = d->workspace_bytes;
  k.N = d->N; k.H = d->H; k.W = d->W; k.C1 = d->C1; k.ldx = d->ldx; k.C2 = d->C2; k.ldx2 = d->ldx2;
  k.Cin = d->C1 + d->C2; k.Cout = d->Cout; k.ldw = d->ldw; k.ldy = d->ldy; k.ldr = d->ldr;
  k.KH = d->KH; k.KW = d->KW; k.stride = d->stride; k.pad_t = d->pad_t; k.pad_l = d->pad_l;
  k.OH = d->OH; k.OW = d->OW; k.OHW = d->OH * d->OW; k.ups = d->upsample2x; k.act = d->act; k.out_f32 = d->out_f32;
  k.n_split = d->yt ? d->n_split : d->Cout; k.t_rows = d->t_rows; k.t_ld = d->t_ld;
  k.out_scale = d->out_scale;
  k.M = d->N * d->OH * d->OW; k.Ktot = d->KH * d->KW * k.Cin; k.nk = (k.Ktot + 63) / 64; k.nbatch = d->nbatch;
  k.bs_x = d->bs_x; k.bs_x2 = d->bs_x2; k.bs_w = d->bs_w; k.bs_bias = d->bs_bias; k.bs_y = d->bs_y; k.bs_r = d->bs_r; k.bias_img = d->bias_img_stride;
  UR_REQUIRE(k.M > 0, "empty problem");
  UR_REQUIRE((long long)d->N * d->H * d->W * (long long)std::max(d->ldx, d->ldx2) < (1ll << 31) &&
                 (long long)d->Cout * d->ldw < (1ll << 31), "tensor too large for 32-bit element offsets");
  { const char* e = getenv("UR_IGEMM_DBG"); k.dbg = e ? atoi(e) : 0; }

  k.kcm = d->k_chunk_major; k.wf = (const uint16_t*)d->w_frag; k.wmajor = 0; k.xgm = 0; k.xbn = 0;
  UR_REQUIRE(!k.kcm || (k.Cin % 64 == 0 && d->C1 % 64 == 0), "k_chunk_major needs C1 and C1+C2 to be multiples of 64");
  k.gn_part = d->gn_part; k.gn_ab = d->gn_ab; k.gn_silu = d->gn_silu; k.f16 = d->dtype == UR_DT_F16; k.gn_fused = 0; k.gn_parts = 0; k.prologue_ok = 0;
  k.dry = dry; k.plan_tn = 0; k.ln_parts = d->ln_parts;
  k.row_stats = d->row_stats; k.ln_stats = d->ln_stats; k.ln_colsum = d->ln_colsum; k.ln_eps = d->ln_eps; k.ln_dim = d->ln_dim;
  UR_REQUIRE(!d->ln_stats || (d->ln_colsum && d->ln_dim > 0), "ln_stats needs ln_colsum / ln_dim");
  // feature combinations the specialised epilogues cover (each is one compiled instance; see epi_frag_pass)
  UR_REQUIRE(!(d->bias_img_stride && (pair || d->ln_stats || d->yt)), "per-image bias rows do not combine with pair activations / LayerNorm folding / transposed columns");
  // staged (LDS-tiled, 16-byte row stores) epilogue: 16-bit y with aligned rows, or no y at all (statistics-only launch)
  k.staged_ok_ = (d->y ? (!d->out_f32 && ((d->ldy | d->bs_y) & 7) == 0) : d->gn_part != nullptr) &&
                 (!d->residual || ((d->ldr | d->bs_r) & 7) == 0);
  UR_REQUIRE(!d->gn_part || (!d->out_f32 && !d->yt), "gn_part needs a plain 16-bit output (or none)");
  hipStream_t s = (hipStream_t)stream;
  const double flops = 2.0 * k.M * (double)k.Cout * k.Ktot * k.nbatch;
  const double bytes = 2.0 * ((double)k.M * k.Cin + (double)k.Cout * k.Ktot + (double)k.M * k.Cout) * k.nbatch;
  const char* fam = d->KH == 3 ? "conv3x3_igemm" : "gemm1x1_igemm";
  static const bool prof_shapes = getenv("UR_PROF_SHAPES") != nullptr;
  if (prof_shapes) {           // per-shape families for offline analysis (interned strings keep the pointers stable)
    static std::map<std::string, int> interned;
    char buf[128];
    snprintf(buf, sizeof buf, "%s M%d N%d K%d s%d u%d b%d a%d", d->KH == 3 ? "c3" : "g1", k.M, k.Cout, k.Ktot, d->stride, d->upsample2x,
             k.nbatch, d->act);
    fam = interned.emplace(buf, 0).first->first.c_str();
  }
  if (dry) {
    const int rc0 = dispatch_conv(k, s, pair);
    if (plan) {
      plan->row_stat_parts = d->row_stats ? k.plan_tn : 0;
      plan->gn_parts = k.gn_parts; plan->gn_fused = k.gn_fused; plan->prologue_ok = k.prologue_ok;
    }
    return rc0;
  }
  if (!d->y && d->gn_part) {            // statistics-only launch: only where the epilogue itself produces the partials
    k.dry = 1;
    const int rc0 = dispatch_conv(k, s, pair);
    if (rc0 != UR_OK) return rc0;
    if (!k.gn_fused || k.splitk > 1) return ur::fail(UR_E_UNSUPPORTED, "ur_conv2d_nhwc: y == NULL needs a launch whose epilogue writes gn_part (see ur_conv2d_plan)");
    k.dry = 0;
  }
  if (d->gn_ab) {
    k.dry = 1;
    const int rc0 = dispatch_conv(k, s, pair);
    if (rc0 != UR_OK) return rc0;
    if (!k.prologue_ok) return ur::fail(UR_E_UNSUPPORTED, "ur_conv2d_nhwc: gn_ab is not supported by this launch (see ur_conv2d_plan.prologue_ok)");
    k.dry = 0;
  }
  ur::ProfScope prof(fam, flops, bytes, s);
  UR_REQUIRE(!(d->row_stats || d->ln_stats) || (k.staged_ok_ && k.nbatch == 1 && !d->yt == !d->yt), "row_stats / ln fusion need a bf16 staged output");
  // Grouped 3x3 convolutions whose groups are halo-kernel sized (chunk-major weights, >= 64 channels in, a multiple of 128 out):
  // one halo launch per group on the channel slice instead of the batched generic kernel, which gathers 64-byte runs per pixel and
  // tap (CFRM's AdaNAFV2.group_conv, densified to 128-channel blocks by the caller: 2.19 ms -> 4 x ~0.2 ms at 256 x 256 x 512)
  static const bool no_ghalo = getenv("UR_IGEMM_NOGHALO") != nullptr;
  if (!no_ghalo && k.nbatch > 1 && d->KH == 3 && k.kcm && k.stride == 1 && !pair && !k.gn_part && !k.row_stats && !k.ln_stats && !k.yt && !k.gn_ab &&
      !k.out_f32 && k.staged_ok_ && k.Cout % 128 == 0 && !k.bias_img && k.C2 == 0) {
    for (int b = 0; b < d->nbatch; ++b) {
      ConvK kb = k;
      kb.nbatch = 1;
      kb.x = k.x + b * k.bs_x; kb.w = k.w + b * k.bs_w; kb.bias = k.bias ? k.bias + b * k.bs_bias : nullptr;
      kb.res = k.res ? k.res + b * k.bs_r : nullptr;
      kb.y = reinterpret_cast<uint16_t*>(k.y) + b * k.bs_y;
      kb.bs_x = kb.bs_w = kb.bs_bias = kb.bs_y = kb.bs_r = 0;
      const int rcb = dispatch_conv(kb, s, pair);
      if (rcb != UR_OK) return rcb;
    }
    return UR_OK;
  }
  const int rc = dispatch_conv(k, s, pair);
  if (rc != UR_OK) return rc;
  if (k.gn_part && !k.gn_fused) {   // this launch could not fuse the statistics: one extra pass over the output
    const int ctot = (pair ? k.Cout / 2 : k.Cout) * k.nbatch;
    UR_REQUIRE(k.y && k.ldy == ctot, "gn_part fallback needs a dense output (ldy == channels)");
    return ur::gn_stats_launch(k.y, k.gn_part, k.N, k.OHW, ctot, d->dtype, s);
  }
  return UR_OK;
}

extern "C" int ur_conv2d_nhwc(const ur_conv_desc* d, ur_stream_t stream) { return conv_impl(d, stream, 0, nullptr); }

// Host-only plan of the launch of `d` (no kernel runs): partial row-sum planes ([parts][M][2] fp32, one per N tile - the
// partial layout keeps the producer free of atomics), GroupNorm partials per image and who writes them, prologue support.
extern "C" int ur_conv2d_plan(const ur_conv_desc* d, ur_conv_plan* plan) {
  UR_REQUIRE(d && plan, "null pointer");
  ur_conv_desc t = *d;
  if (!t.y && !t.yt && !t.gn_part) t.y = reinterpret_cast<void*>(16);    // any non-null value: only the plan is wanted
  return conv_impl(&t, nullptr, 1, plan);
}

extern "C" int ur_gemm_bias_act(const void* x, const void* w, const float* bias, const void* residual, void* y, long long M, int N, int K,
                                int ldx, int ldw, int ldy, int ldr, int act, float* workspace, size_t workspace_bytes, int dtype,
                                ur_stream_t stream) {
  UR_REQUIRE(M > 0 && M < (1ll << 31), "M out of range");
  ur_conv_desc d = {};
  d.x = x; d.w = w; d.bias = bias; d.residual = residual; d.y = y;
  d.workspace = workspace; d.workspace_bytes = workspace_bytes;
  d.N = 1; d.H = 1; d.W = (int)M; d.C1 = K; d.ldx = ldx; d.Cout = N; d.ldw = ldw; d.ldy = ldy; d.ldr = ldr;
  d.KH = d.KW = 1; d.stride = 1; d.OH = 1; d.OW = (int)M; d.act = act; d.out_scale = 1.f; d.nbatch = 1; d.dtype = dtype;
  return conv_impl(&d, stream, 0, nullptr);
}

extern "C" int ur_groupconv3x3_nhwc(const void* x, const void* w, const float* bias, void* y, int N, int H, int W, int Cg, int Cog,
                                    int groups, int act, float* workspace, size_t workspace_bytes, int dtype, ur_stream_t stream) {
  UR_REQUIRE(groups >= 1 && Cg % 8 == 0 && Cog % 4 == 0, "groups >= 1, Cg % 8 == 0, Cog % 4 == 0");
  ur_conv_desc d = {};
  d.x = x; d.w = w; d.bias = bias; d.y = y;
  d.workspace = workspace; d.workspace_bytes = workspace_bytes;
  d.N = N; d.H = H; d.W = W; d.C1 = Cg; d.ldx = Cg * groups; d.Cout = Cog; d.ldw = 9 * Cg; d.ldy = Cog * groups;
  d.KH = d.KW = 3; d.stride = 1; d.pad_t = d.pad_l = 1; d.OH = H; d.OW = W; d.act = act; d.out_scale = 1.f; d.nbatch = groups;
  d.bs_x = Cg; d.bs_w = (long long)Cog * 9 * Cg; d.bs_bias = Cog; d.bs_y = Cog; d.dtype = dtype;
  return conv_impl(&d, stream, 0, nullptr);
}
