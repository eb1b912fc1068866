/*
 * unirestore_hip.h — C ABI of libunirestore_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for UniRestore's diffusion-prior restoration hot path
 * (DiffUIE.forward, /root/reference/src/modules/diffuie/unifie.py:107-169).  The reference has no
 * native layer: every entry point below replaces an ATen/cuDNN/cuBLAS op that the reference reaches
 * through torch.nn / diffusers.  The "replaces" notes cite the reference call sites.
 *
 * Conventions
 *   - every pointer is a DEVICE address unless marked "host"; activations are NHWC 16-bit (raw uint16 bit
 *     patterns of bf16 or fp16, chosen per call by a UR_DT_* `dtype`), statistics / tables / tiny vectors are fp32;
 *   - no kernel uses atomics: two runs (or two hipGraph replays) on the same inputs are bit-identical;
 *   - the library never allocates: workspaces are passed in by the caller;
 *   - every call is asynchronous on `stream` (a hipStream_t) and safe under hipGraph capture;
 *   - return value: 0 = UR_OK, negative = UR_E_*; ur_last_error() gives the message (thread-local).
 */
#ifndef UNIRESTORE_HIP_H
#define UNIRESTORE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ur_stream_t; /* hipStream_t */

enum { UR_OK = 0, UR_E_INVALID = -1, UR_E_UNSUPPORTED = -2, UR_E_LAUNCH = -3, UR_E_WORKSPACE = -4 };

/* 16-bit storage / matrix-core operand type of activations and weights (accumulation, statistics, softmax, the DDIM state
 * and every tiny vector are fp32 in both).  UR_DT_BF16: precision: bf16-mixed of the reference (configs/val.yaml:12);
 * UR_DT_F16: IEEE half - same bytes and MFMA rate, 8x smaller rounding error per stored tensor; a value beyond +-65504
 * overflows to +-inf (IEEE; rounds 2-4 clamped) and turns into NaN in the next normalisation / softmax, so an fp16 range
 * problem shows up in the output instead of as a silently clipped activation - fall back to UR_DT_BF16 for such weights
 * (BASELINE.json configs[4], "fp16"). */
enum { UR_DT_BF16 = 0, UR_DT_F16 = 1 };

/* epilogue activations */
enum {
  UR_ACT_NONE = 0,
  UR_ACT_SILU = 1,
  UR_ACT_GELU = 2,  /* exact erf GELU (nn.GELU default) */
  UR_ACT_GEGLU = 3, /* out[j] = a[j] * gelu(g[j]); weight rows pre-interleaved in blocks of 32 (a|g) */
  UR_ACT_GATE = 4,  /* NAFNet SimpleGate: out[j] = a[j] * g[j]; same interleave */
  UR_ACT_TANH = 5,
  UR_ACT_RELU = 6   /* SPADE's mlp_shared (spade.py:47-49) */
};

int ur_version(void);
const char* ur_last_error(void);

/* ---- implicit-GEMM convolution / linear (bf16 MFMA, fp32 accumulate) -------------------------
 * y[n,oh,ow,co] = epilogue( sum_{kh,kw,ci} x[n, ih, iw, ci] * w[co,kh,kw,ci] )
 * Replaces: nn.Conv2d 3x3/1x1 (ResnetBlock2D conv1/conv2/conv_shortcut, conv_in/out, Downsample2D,
 *   Upsample2D.conv — base_model.py:94-209, controller.py:193-220, autoencoder.py:11-72), nn.Linear
 *   (Transformer2DModel / BasicTransformerBlock / Attention projections), CSCEAdapter 1x1 convs
 *   (scedit.py:28-38), NAFBlock/AdaNAFV2 1x1 + grouped convs (nafnet_arch.py:32-95, cfrm.py:18-36),
 *   TaskFeatureAdapter convs (taskeditor.py:20-52), torch.cat on the UNet up path (base_model.py:189,197;
 *   "virtual concat": x2), F.interpolate(nearest, 2x) in Upsample2D (upsample2x).
 * Epilogue order: acc (+bias[co]) -> act -> (*out_scale) -> (+residual) -> store.
 * Batched / grouped form: blockIdx.y = b in [0,nbatch): every base pointer advances by its bs_* stride.
 */
typedef struct ur_conv_desc {
  const void* x;        /* bf16 [N,H,W,ldx]; first C1 input channels */
  const void* x2;       /* bf16 [N,H,W,ldx2] or NULL; next C2 input channels (virtual concat) */
  const void* w;        /* bf16 [Cout][KH*KW*(C1+C2)] row stride ldw */
  const float* bias;    /* fp32 [Cout] or NULL */
  const void* residual; /* bf16 [M, ldr] or NULL (M = N*OH*OW) */
  void* y;              /* 16-bit (or fp32 if out_f32) [M, ldy]; may be NULL if only gn_part is wanted */
  void* yt;             /* bf16 transposed output for columns >= n_split: [M/t_rows][Cout-n_split][t_ld] or NULL */
  float* gn_part;       /* fp32 [N][P][nbatch*Cout_out][2] = partial (sum, sum of squares) of the 16-bit outputs per image,
                           row chunk and channel, plain stores (P = ur_conv_plan.gn_parts): statistics for the GroupNorm /
                           InstanceNorm / AdaptiveAvgPool2d(1) that consumes y (ur_groupnorm_finalize), or NULL.  With
                           ur_conv_plan.gn_fused the epilogue writes them and y may be NULL (pooled output only:
                           taskeditor.py:35); otherwise one extra pass over y does */
  const float* gn_ab;   /* fp32 [N][2][C1+C2] per-(image, input channel) affine (a | b) from ur_groupnorm_finalize, or NULL:
                           the conv reads act(a*x + b) instead of x (GroupNorm apply [+ SiLU] of ResnetBlock2D fused into the
                           conv's loader; zero padding stays zero).  Only where ur_conv_plan.prologue_ok */
  int gn_silu;          /* 1: SiLU after the gn_ab affine */
  float* row_stats;     /* fp32 [parts][M][2] = per-row (sum, sum of squares) of this GEMM's output, one partial plane per N
                           tile (plain stores; parts = ur_conv_plan.row_stat_parts from ur_conv2d_plan): the LayerNorm statistics of the
                           GEMM that consumes y (BasicTransformerBlock norm1-3) or NULL */
  const float* ln_stats;  /* fp32 [ln_parts][M][2] row sums of x (a producer's row_stats): this GEMM computes W.LayerNorm(x) as
                             rstd*(acc - mean*ln_colsum[n]) + bias[n] with w = W*gamma, bias = W.beta + b prefolded; or NULL */
  const float* ln_colsum; /* fp32 [Cout]: sum_k of the (bf16) folded weight row */
  float ln_eps;
  int ln_dim;             /* normalised width (= K) */
  int ln_parts;           /* number of partial planes in ln_stats */
  float* workspace;     /* fp32 split-K scratch or NULL */
  size_t workspace_bytes;
  int N, H, W;          /* input dims (before upsample2x) */
  int C1, ldx, C2, ldx2;
  int Cout, ldw, ldy, ldr;
  int KH, KW, stride, pad_t, pad_l;
  int OH, OW;
  int upsample2x;       /* 1: x is read through a nearest-neighbour 2x upsample */
  int act;              /* UR_ACT_* */
  int out_f32;
  int n_split, t_rows, t_ld;
  float out_scale;      /* applied after act; 1.0 for none */
  int nbatch;           /* >= 1 */
  long long bs_x, bs_x2, bs_w, bs_bias, bs_y, bs_r; /* element strides per batch index */
  int k_chunk_major;    /* 1: weight K index runs (64-channel chunk, tap, channel) instead of (tap, channel): the taps of
                           one chunk are consecutive K tiles, so re-reads of a pixel hit L2 (needs C1, C1+C2 % 64 == 0) */
  long long bias_img_stride; /* 0: one bias row; else bias row of image n = m/(OH*OW) is bias + n*stride
                                (per-sample time embeddings, unifie.py:91-105) */
  int dtype;            /* UR_DT_BF16 | UR_DT_F16: type of x, x2, w, residual, y (unless out_f32), yt */
  const void* w_frag;   /* optional second packing of the SAME weights for the weight-streaming kernel of the 8 x 8 maps (3x3, stride 1,
                           Cout % 128 == 0, (C1+C2) % 256 == 0), or NULL: MFMA-fragment-major
                           [Cout/128][(C1+C2)/64][tap 9][k-step 4][row block 4][lane 64][8], lane = (cout % 32) + 32 * ((cin % 16) / 8),
                           element = cin % 8 - one coalesced 1-KiB load per 32 x 16 weight fragment (csrc/conv_wstream.hip) */
} ur_conv_desc;

int ur_conv2d_nhwc(const ur_conv_desc* d, ur_stream_t stream);

/* host-only query (no launch): what the launch of `d` will do.  Fill d exactly as for the real call (non-NULL dummies
 * are fine for row_stats / gn_part / gn_ab when only the plan is wanted). */
typedef struct ur_conv_plan {
  int row_stat_parts; /* partial planes the launch writes into d->row_stats */
  int gn_parts;       /* partials per image (P) the launch writes into d->gn_part */
  int gn_fused;       /* 1: written by the conv epilogue itself (y may be NULL); 0: by an extra pass over y */
  int prologue_ok;    /* 1: d->gn_ab is honoured inside this launch; 0: apply the GroupNorm separately */
} ur_conv_plan;
int ur_conv2d_plan(const ur_conv_desc* d, ur_conv_plan* plan);

/* Linear / 1x1 convolution with the epilogues of SURVEY.md 8(b): y[m, n] = act(x[m, :] . w[n, :] + bias[n]) (+ residual);
 * act in UR_ACT_* (GEGLU / GATE: w rows pre-interleaved, N = 2 * output columns).  Thin wrapper over ur_conv2d_nhwc. */
int ur_gemm_bias_act(const void* x, const void* w, const float* bias, const void* residual, void* y, long long M, int N, int K,
                     int ldx, int ldw, int ldy, int ldr, int act, float* workspace, size_t workspace_bytes, int dtype,
                     ur_stream_t stream);
/* grouped 3x3 convolution (pad 1, stride 1) + activation: AdaNAFV2.group_conv (cfrm.py:20-21), the three TFA gate branches
 * (taskeditor.py:30-52).  x [N,H,W,G*Cg], w [G*Cog][9*Cg], y [N,H,W,G*Cog].  Thin wrapper over ur_conv2d_nhwc (nbatch = G). */
int ur_groupconv3x3_nhwc(const void* x, const void* w, const float* bias, void* y, int N, int H, int W, int Cg, int Cog,
                         int groups, int act, float* workspace, size_t workspace_bytes, int dtype, ur_stream_t stream);

/* ---- normalisation (HBM-bound) ----------------------------------------------------------------
 * GroupNorm over NHWC (+ optional SiLU) in three steps, all without atomics:
 *   1. statistics: fp32 partial planes part[N][P][C][2] (sum, sum of squares per image, pixel chunk, channel) - written by the
 *      PRODUCER's epilogue (ur_conv_desc.gn_part) or by ur_groupnorm_stats;
 *   2. ur_groupnorm_finalize: partials of one or two (virtually concatenated) sources -> fp32 ab[N][2][C] with
 *      y = a*x + b == GroupNorm(x); fp64 sums in a fixed order;
 *   3. ur_groupnorm_apply_act (or ur_conv_desc.gn_ab: applied inside the consuming convolution).
 * G == C with gamma = beta = NULL gives InstanceNorm2d; mean_out gives AdaptiveAvgPool2d(1).
 * Replaces: nn.GroupNorm(32,C)+SiLU in every ResnetBlock2D / conv_norm_out, Transformer2D / Attention
 *   pre-norms, AdaNAFV2.group_norm (cfrm.py:19), nn.InstanceNorm2d (taskeditor.py:31,40,49), nn.AdaptiveAvgPool2d(1).
 */
int ur_groupnorm_stats_parts(int N, int HW, int C);   /* host: P of the plane ur_groupnorm_stats writes */
size_t ur_groupnorm_ws_bytes(int N, int HW, int C);   /* bytes of that plane */
size_t ur_groupnorm_ab_bytes(int N, int C);           /* bytes of an ab table */
int ur_groupnorm_stats(const void* x, float* part, int N, int HW, int C, int dtype, ur_stream_t stream);
int ur_instnorm_stats(const void* x, float* part, int N, int HW, int C, int dtype, ur_stream_t stream); /* same pass */
/* part2/parts2/C2 (optional): second source, virtually concatenated after part1's C1 channels (UNet up path: GroupNorm over
 * torch.cat([sample, skip]), base_model.py:189,197).  ab and/or mean_out ([N][G] group means) may be NULL. */
int ur_groupnorm_finalize(const float* part1, int parts1, int C1, const float* part2, int parts2, int C2, const float* gamma,
                          const float* beta, int N, int HW, int G, float eps, float* ab, float* mean_out, ur_stream_t stream);
/* y[N,HW,C1+C2] = act(a*x + b) over x [N,HW,C1] (| x2 [N,HW,C2]) */
int ur_groupnorm_apply_act(const void* x, const void* x2, void* y, const float* ab, int N, int HW, int C1, int C2, int silu,
                           int dtype, ur_stream_t stream);
/* the three chained.  pre1 / pre2 (optional): producer-side partial planes of x / x2 with parts1 / parts2 partials per image;
 * ws: ur_groupnorm_ws_bytes(N,HW,C1) [+ (N,HW,C2)] bytes of scratch, needed only for sources without producer partials */
int ur_groupnorm_nhwc(const void* x, const void* x2, void* y, const float* gamma, const float* beta, int N, int HW,
                      int C1, int C2, int G, float eps, int silu, float* ws, float* ab, const float* pre1, int parts1,
                      const float* pre2, int parts2, int dtype, ur_stream_t stream);
/* LayerNorm over the last dim of [rows, C] (nn.LayerNorm in BasicTransformerBlock; timm LayerNorm2d
 * in NAFBlock, nafnet_arch.py:97-98, which is LayerNorm-over-C in NHWC). */
int ur_layernorm_rows(const void* x, void* y, const float* gamma, const float* beta, long long rows, int C,
                      float eps, int dtype, ur_stream_t stream);
/* softmax over rows of an fp32 [rows, cols] matrix -> 16-bit (upcast_softmax). */
int ur_softmax_rows_f32(const float* s, void* p, long long rows, int cols, int ldp, int dtype, ur_stream_t stream);

/* ---- attention (flash-style, bf16 MFMA, fp32 softmax) ------------------------------------------
 * o[b,t,h*D+d] = softmax_k(q.k * scale) v.  q:[B][Tq][ldq], k:[B][Tk][ldk] (head h at column h*D),
 * vt:[B][H*D][ldvt] (V transposed: row = channel, column = key index), o:[B][Tq][ldo].  D in {64,128,512}
 * (512: the single 512-wide head of the VAE mid-block attention, split over keys for q.k and over channels for p.v
 *  inside one workgroup - no Tq x Tk matrix is materialised at any D).
 * Scale convention for D == 64: the kernels work in the exp2 domain on scores multiplied by scale * log2(e).  Callers that
 *   want full accuracy fold that factor into the QUERY PROJECTION's fp32 weights before their one rounding to 16 bits and pass
 *   scale = ln 2 (scale * log2(e) == 1: q is used as it is) - unirestore_amd/modules/nn.py does.  Any other scale is honoured,
 *   but the ping-pong kernel then multiplies the 16-bit q by the factor and rounds it to 16 bits a second time (bf16: one more
 *   2^-9 relative rounding per q element; on scores of a few hundred that moves single softmax weights by ~5 %).
 * Replaces: F.scaled_dot_product_attention via diffusers AttnProcessor2_0 (UNet self/cross attention,
 *   Controller AttnDownBlock2D / UNetMidBlock2D attention) and the VAE mid-block AttentionBlock
 *   (/root/reference/src/modules/diffuie/autoencoder.py:37-45 calls it through vae.encoder/decoder.mid_block).
 */
int ur_attention_fwd(const void* q, const void* k, const void* vt, void* o, int B, int H, int Tq, int Tk,
                     int D, int ldq, int ldk, int ldvt, int ldo, long long bs_q, long long bs_k,
                     long long bs_vt, long long bs_o, float scale, int dtype, ur_stream_t stream);
/* The same with a scratch buffer.  The d = 64 self-attention kernel runs one 256-query workgroup per CU; when the launch is
 * whole rounds of the 256 CUs plus at most half a round (B = 8, 5 heads, T = 4096: 640 workgroups), the query tiles of that
 * remainder are split in two key halves that fill the last round, and a second small kernel merges the halves from `ws`
 * (un-normalised fp32 rows + running maximum + row sum).  ur_attention_workspace_bytes() is what the split needs (0: none);
 * ws == NULL or a smaller buffer just runs unsplit.  The library never allocates: the caller owns ws (16-byte aligned). */
size_t ur_attention_workspace_bytes(int B, int H, int Tq, int Tk, int D);
int ur_attention_fwd_ws(const void* q, const void* k, const void* vt, void* o, int B, int H, int Tq, int Tk,
                        int D, int ldq, int ldk, int ldvt, int ldo, long long bs_q, long long bs_k, long long bs_vt,
                        long long bs_o, float scale, void* ws, size_t ws_bytes, int dtype, ur_stream_t stream);

/* ---- token-stationary fused chains (csrc/tchain.hip) ----------------------------------------------
 * A wave keeps 32 tokens in registers through a whole chain of token-wise layers; the weights arrive as a pre-packed
 * stream of ur_chain_tile_bytes()-byte tiles that are the LDS image of each layer (unirestore_amd/chain.py packs them:
 * [row][128 B] blocks, XOR-swizzled slots, swap23-permuted output rows, LayerNorm folded, fp32 vectors in the tile tail).
 * Shapes: C == 320 (UNet level 0) and T % 128 == 0; anything else returns UR_E_UNSUPPORTED / UR_E_INVALID and the caller
 * uses the per-layer entry points above. */
size_t ur_chain_tile_bytes(void);
/* y[t,:] = x[t,:] + W2 . GEGLU(W1 . LayerNorm(x[t,:]) + b1) + b2: diffusers FeedForward(activation_fn="geglu") + norm3 + residual of
 * BasicTransformerBlock (reached through /root/reference/src/modules/diffuie/base_model.py:137-160,184-198).  The 4C-wide
 * hidden tensor never exists in memory.  stream_w: 3 * hidden/64 tiles. */
int ur_ff_geglu_fused(const void* x, const void* stream_w, size_t stream_bytes, void* y, long long T, int C, int hidden, int ldx,
                      int ldy, float ln_eps, int dtype, ur_stream_t stream);

/* h0 = proj_in(GroupNorm(x)) (gn_ab: the per-image affine [N][2][C] from ur_groupnorm_finalize, applied to the fragments in
 * registers) and q | k | v = to_q / to_k / to_v(LayerNorm1(h0)) of the block's self-attention: Transformer2DModel.norm + proj_in,
 * BasicTransformerBlock.norm1 + attn1 projections (base_model.py:137-160 via diffusers).  h0, q, k: [T][C]; vt: [N][C][tokens_per_image]
 * (V transposed, the layout ur_attention_fwd reads).  stream_w: 20 tiles. */
int ur_transformer_head_fused(const void* x, const float* gn_ab, const void* stream_w, size_t stream_bytes, void* h0, void* q, void* k,
                              void* vt, long long T, int tokens_per_image, int C, float ln_eps, int dtype, ur_stream_t stream);
/* Everything behind the self-attention: h1 = h0 + to_out(o1); cross-attention over the CONSTANT context (its K / V are baked into
 * the stream; tk <= 80 keys, heads x 64) with LayerNorm2 folded into to_q; h2 = h1 + to_out(o2); h3 = h2 + FF(LayerNorm3(h2));
 * y = xres + proj_out(h3).  gn_part (optional): fp32 [N][tokens_per_image/128][C][2] partial (sum, sum of squares) of y for the
 * GroupNorm of the next ResnetBlock2D (ur_groupnorm_finalize with parts = tokens_per_image/128).  stream_w: 25 + 3*hidden/64 tiles. */
int ur_transformer_tail_fused(const void* o1, const void* h0, const void* xres, const void* stream_w, size_t stream_bytes, void* y,
                              float* gn_part, long long T, int tokens_per_image, int C, int hidden, int heads, int tk, float ln_eps,
                              float attn_scale, int dtype, ur_stream_t stream);

/* SC-Tuner adapter (CSCEAdapter, /root/reference/src/modules/diffuie/scedit.py:24-38) in one launch: s = x + proj(cond),
 * y = tuner.2(GELU(tuner.0(s))) + s for a 320-channel UNet skip x [T][320] and the 256-channel Controller feature cond [T][256];
 * gn_part (optional) as in ur_transformer_tail_fused (the edited skip feeds a GroupNorm on the up path).  stream_w: 14 tiles. */
int ur_csce_fused(const void* x, const void* cond, const void* stream_w, size_t stream_bytes, void* y, float* gn_part, long long T,
                  int tokens_per_image, int C, int Ccond, int dtype, ur_stream_t stream);

/* ---- HBM-bound stencils / reductions / elementwise ---------------------------------------------*/
/* depthwise 3x3 (pad 1) + bias, optional SimpleGate (out channels C/2): nafnet_arch.py:41-49,22-25 */
int ur_dwconv3x3_nhwc(const void* x, const float* w9c, const float* bias, void* y, int N, int H, int W, int C,
                      int gate, int dtype, ur_stream_t stream);
/* mean over HW -> fp32 [N][C] (nn.AdaptiveAvgPool2d(1)): statistics pass + finalize; ws = ur_groupnorm_ws_bytes(N,HW,C) bytes */
int ur_avgpool_hw(const void* x, float* out, int N, int HW, int C, float* ws, int dtype, ur_stream_t stream);
/* y = x * s[n][c] (+ residual) : channel attention scaling (nafnet_arch.py:116, cfrm.py:46-48, taskeditor.py:95) */
int ur_scale_channels(const void* x, const float* s, const void* residual, void* y, int N, int HW, int C,
                      int dtype, ur_stream_t stream);
/* y = a + b * s[c] (per-channel learnable residual scale beta/gamma, nafnet_arch.py:121,130) */
int ur_axpy_channels(const void* a, const void* b, const float* s, void* y, long long rows, int C, int dtype, ur_stream_t stream);
/* SPADE modulation (spade.py:69): y = n * (1 + gamma) + beta (+ residual); gamma | beta = gb[:, 0:C] | gb[:, C:2C] (row stride ldgb) */
int ur_spade_modulate(const void* n, const void* gb, int ldgb, const void* residual, void* y, long long rows, int C,
                      int dtype, ur_stream_t stream);
/* tiny fp32 linear: y[m, g*Ng+n] = act(bias + sum_k x[m, g*Kg+k] * w[g*Ng+n, k]) (time MLPs, SCA, gates) */
int ur_linear_f32(const float* x, const float* w, const float* bias, float* y, int M, int N, int K, int groups,
                  int act, ur_stream_t stream);
/* TFA prompt update (taskeditor.py:80-91): pooled [B][3][T*D] (filter, info, content), cond [B][T][D] -> upd */
int ur_tfa_prompt_update(const float* pooled, const float* cond, float* upd, int B, int T, int D, ur_stream_t stream);
/* out[n][c] = a[n][c] * b[n][c / (C/G)]  (combine intra/inter group attention, cfrm.py:46-48) */
int ur_vec_mul_group(const float* a, const float* b, float* out, int N, int C, int G, ur_stream_t stream);

/* ---- latent / image boundary --------------------------------------------------------------------*/
/* images NCHW fp32 in [0,1] -> NHWC 16-bit (x*2-1), channels padded with zeros to Cpad (autoencoder.py:149) */
int ur_image_to_nhwc(const float* img, void* y, int N, int C, int H, int W, int Cpad, int dtype, ur_stream_t stream);
/* NHWC fp32/16-bit [N,H,W,ld] first C channels -> NCHW fp32, out = x*mul+add (autoencoder.py:175) */
int ur_nhwc_to_nchw_f32(const void* x, int x_is_f32, float* out, int N, int C, int H, int W, int ld, float mul,
                        float add, int dtype, ur_stream_t stream);
/* NCHW fp32 -> NHWC 16-bit with channel padding (module-level API plumbing) */
int ur_nchw_f32_to_nhwc(const float* x, void* y, int N, int C, int H, int W, int Cpad, int dtype, ur_stream_t stream);
/* DiffUIE.forward pre-processing (reference unifie.py:124-134) fused with the encoder's x*2-1 (autoencoder.py:152 -> vae.encode)
 * and the NHWC layout pass: img fp32 [N,C,H,W] -> F.interpolate(bicubic, align_corners=False, antialias=False) to RH x RW
 * (skipped when equal) -> F.pad(reflect) right/bottom by PW/PH -> v*mul+add -> y 16-bit [N,RH+PH,RW+PW,Cpad]. */
int ur_image_resize_pad_nhwc(const float* img, void* y, int N, int C, int H, int W, int RH, int RW, int PH, int PW, int Cpad,
                             float mul, float add, int dtype, ur_stream_t stream);
/* DiffUIE.forward post-processing (unifie.py:164-168) and the evaluator's 8-bit quantisation (eval_image_restoration.py:71):
 * x NHWC (16-bit | fp32) [N,XH,XW,ld] -> v*mul+add -> crop [0:CH,0:CW] -> bicubic to OH x OW -> optional
 * mul(255).round().clamp(0,255).div(255) -> out fp32 [N,C,OH,OW]. */
int ur_image_unpad_resize_nchw(const void* x, int x_is_f32, float* out, int N, int C, int XH, int XW, int ld, int CH, int CW,
                               int OH, int OW, float mul, float add, int quantize, int dtype, ur_stream_t stream);
/* z = (mean + exp(0.5*clamp(logvar,-30,20)) * noise) * scale; moments NHWC fp32 [M, ld] (mean | logvar) */
int ur_vae_sample(const float* moments, int ld, const float* noise_nchw, float* z_nhwc, void* z_16, int N,
                  int HW, int Clat, int Cpad, float scale, int dtype, ur_stream_t stream);
/* zt = sa * z0 + sb * noise (DDPMScheduler.add_noise, unifie.py:88); fp32 NHWC state + 16-bit copy */
int ur_add_noise(const float* z0, const float* noise_nchw, float* zt, void* zt_16, int N, int HW, int Clat,
                 int Cpad, float sa, float sb, int dtype, ur_stream_t stream);
/* DDIM update (unifie.py:150): zt <- c_x*zt + c_e*eps, eps fp32 NHWC [M, ld_eps]; refreshes the 16-bit copy */
int ur_ddim_step(float* zt, const float* eps, int ld_eps, void* zt_16, long long M, int Clat, int Cpad,
                 float c_x, float c_e, int dtype, ur_stream_t stream);
/* y_16[M][Cpad] = x_f32[M][ld] * mul (latents / scaling_factor before post_quant_conv) */
int ur_f32_to_bf16_scaled(const float* x, int ld, void* y, long long M, int C, int Cpad, float mul, int dtype, ur_stream_t stream);

/* ---- live per-kernel-family timing (HIP events on the launch stream) ------------------------------*/
int ur_profile_enable(int on);
/* writes a JSON object {family: {launches, ms, flops, bytes}} into buf (host); synchronises the events */
int ur_profile_report(char* buf, size_t buf_bytes);

#ifdef __cplusplus
}
#endif
#endif /* UNIRESTORE_HIP_H */
