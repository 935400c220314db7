/* unidepth_hip.h -- C-ABI of the MI355X (gfx950) kernel library behind UniDepthV2.infer().
 *
 * The reference has no FFI on this path: every op below replaces a torch call made by the reference's
 * Python modules (file:line relative to the reference tree).  Conventions mirror the reference's own
 * native-op convention (unidepth/ops/extract_patches/src/extract_patches.cpp:3-6, ops/knn/src/knn.cu:130,330-341)
 * minus the torch types: raw device pointers + sizes, the caller owns every buffer (incl. workspace),
 * work is enqueued on the hipStream_t passed in (`void* stream`; NULL = default stream), no hidden
 * allocation, no host sync.  Return value: 0 = OK, negative = UD_ERR_* (the Python side raises RuntimeError).
 * All functions are stateless and re-entrant; one process per GPU.
 *
 * Activations are fp16 (MFMA operands) or fp32 (residual streams, statistics, outputs); weights are fp16,
 * [N, K] row-major with K contiguous and K padded to a multiple of 64 (zeros).
 */
#ifndef UNIDEPTH_HIP_H
#define UNIDEPTH_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UD_OK 0
#define UD_ERR_BAD_ARG (-1)
#define UD_ERR_LAUNCH (-2)
#define UD_ERR_UNSUPPORTED (-3)

/* ---- GEMM epilogues ------------------------------------------------------------------------------- */
#define UD_EPI_F16 0   /* out(fp16)[row, n] = act(acc + bias[n] + add[.., n])                                      */
#define UD_EPI_F32 1   /* out(fp32)[row, n] (+)= acc + bias[n] + add[.., n], then `act` (NONE / GELU / CLAMPEXP) on the stored value;
                        * optional fp16 copy out2 = act2(result)  */
#define UD_EPI_QKV 2   /* n <  vsplit: out(fp16) row-major;  n >= vsplit: out2(fp16) = V^T [img][head][64][kv_ld], the 4-key blocks
                        * of every aligned 16-key group stored in the order [0, 2, 1, 3] (= the k-slot order of the P V MFMA,
                        * so the attention kernel DMAs V^T tiles to LDS as they are): key t lives at column
                        * (t & ~15) | ((t & 4) << 1) | ((t & 8) >> 1) | (t & 3)                                                       */
#define UD_EPI_D2S 3   /* ConvTranspose(k=s) depth-to-space: out(fp32 NHWC) += acc + bias[o]; out2 fp16 = act2(..)   */
#define UD_EPI_HEAD 4  /* y = sum_n lrelu(acc + bias[n]) * w2[n] + b2;  out(fp32)[m] = exp(clip(y,-8,8) + post_add) */

#define UD_ACT_NONE 0
#define UD_ACT_GELU 1   /* exact erf GELU (nn.GELU default; reference metadinov2/mlp.py:35-41, layers/mlp.py:27) */
#define UD_ACT_LRELU 2  /* LeakyReLU(0.01) (reference layers/upsample.py:164) */
#define UD_ACT_CLAMPEXP 3  /* exp(clamp(x, -10, 10)) (UniDepthV1 multi-scale outputs, unidepthv1/decoder.py:322-325); UD_EPI_F32 `act` only */

#define UD_A_DENSE 0          /* A is [M, lda] row-major                                              */
#define UD_A_CONV3_ZERO 1     /* A rows are gathered 3x3 taps of an NHWC image, zero padding           */
#define UD_A_CONV3_REFLECT 2  /* same, reflect padding (reference decoder.py:199-226)                   */
#define UD_A_CONV3_REFLECT_UP 3  /* reflect-padded 3x3 taps of the bilinear align_corners=True up-sampling of A [B, Hsrc, Wsrc, Cin] to
                                  * (Himg, Wimg) (decoder.py:299-301,309-311 fused into the following conv; UD_EPI_HEAD only); img_stride and gA
                                  * describe the LOW-resolution A */

/* C[M,N] = A[M,K] * W[N,K]^T on v_mfma_f32_16x16x32_f16 tiles, fp32 accumulate.
 * Replaces: nn.Linear / F.linear everywhere on the path (metadinov2/attention.py:53,60; mlp.py:36-40;
 * layers/attention.py:117-122,139; layers/mlp.py:30-33; decoder.py:44,145), the patch-embed Conv2d
 * (metadinov2/patch_embed.py:66-88), ConvTranspose2d (decoder.py:166-173), 3x3/1x1 Conv2d
 * (layers/upsample.py:148-163,208-214; decoder.py:199-226) as implicit GEMMs. */
typedef struct UdGemm {
  const void* A;
  const void* W;
  const float* bias;      /* [N] fp32 or NULL */
  void* out;
  void* out2;
  const float* add;       /* optional fp32 [*, ldadd] added before activation; row = (m % rows_in) + add_row_off */
  const void* zeros;      /* >= 256 B of zeros in device memory (padding source for conv A-modes) */
  const float* w2;        /* UD_EPI_HEAD: [N] second-layer (1x1 conv) weights */
  int M, N, K;            /* K = padded reduction length (multiple of 64) */
  int lda, ldw, ldc, ldc2, ldadd;
  int amode, epi, act, act2;
  int accumulate;         /* UD_EPI_F32: 0 = overwrite, 1 = out += result, 2 = result = out + acc but only the fp16 copy out2 is written */
  int rows_in, rows_out, row_off, add_row_off;   /* output row = (m / rows_in) * rows_out + (m % rows_in) + row_off; rows_in = 0 -> identity */
  /* conv A-modes: row m -> image m / rows_img, pixel p = m % rows_img (valid if p < Himg*Wimg) */
  int Himg, Wimg, Cin, cstride, coff, rows_img;
  long long img_stride;   /* elements between images of A */
  /* UD_EPI_QKV */
  int vsplit, tok_per_img, kv_ld, heads_v;
  /* UD_EPI_D2S: n = (a*k + c)*Co + o ; input row m -> image m / rows_in_img, p = m % rows_in_img -> (y, x) on [Hin, Win] */
  int d2s_k, d2s_Co, d2s_Hin, d2s_Win, d2s_rows_in_img;
  long long d2s_out_img_pix;   /* pixels between images in the output */
  /* UD_EPI_HEAD */
  float b2, post_add;
  /* groups: blockIdx.z = g offsets (elements) */
  int groups;
  long long gA, gW, gBias, gOut, gOut2, gW2;
  float b2_g1, post_add_g1;    /* group 1 constants for UD_EPI_HEAD */
  int tile_hint;               /* 0 = auto, 1 = force 128x128 tiles, 2 = force 256x256, 3 = force 192x256 (dense A only),
                                  5 / 6 = 128x128 tiles: plain 2-stage kernel / 4-stage pipelined ring (6 is what auto picks when the
                                  tile count is at most the CU count and K >= 512), 7 = 6 + the two-way K split below,
                                  8 = row-balanced schedule of the 256-column kernel (dense A; auto picks it when the tile list
                                  would leave the last round partly empty), 9 = 192x256 tiles with the 2-deep weight ring (dense
                                  192-row launches otherwise fetch the weight operand two K-tiles ahead through a 3-deep LDS ring:
                                  same bits, the weights of a layer are cold in every step),
                                  10 = 192x256 tiles with the two-way K split of the large-tile list (splitk_ws_bytes below; refused -> 128x128 tiles) */
  /* optional scratch for the two-way K split of small problems (at most 128 tiles of 128x128, K >= 1024, dense A, F16 / F32
   * epilogues): two workgroups on different CUs each take half of K, the later one adds the other's fp32 partial tile (a + b is
   * order-independent, so results do not depend on timing) and runs the epilogue.  splitk_ws: 2 * tiles * 64 KB; splitk_cnt: one
   * unsigned per tile, zeroed ONCE by the caller (tickets are used by parity and never reset).  Both owned by the caller and not
   * shared between streams.  NULL = never split. */
  void* splitk_ws;
  void* splitk_cnt;
  int Hsrc, Wsrc;              /* UD_A_CONV3_REFLECT_UP: size of the low-resolution source image */
  /* Split-fp16 products by K concatenation (no second kernel): the reduction index may run PAST the stored width of one operand and
   * wraps around once.  Dense A: a_wrap = stored columns of A (multiple of 64), column k of the product reads A[:, k - a_wrap] for
   * k >= a_wrap; conv A-modes: a_wrap = stored channels per pixel (multiple of 8), `Cin` = channels the K index is decoded with
   * (e.g. 2 * a_wrap), channel c >= a_wrap reads channel c - a_wrap.  w_wrap: the same for W (dense A only).  0 = no wrap.
   * Use: W' = [W_hi | W_lo] (W_lo = fp16(W - W_hi), packed once at load time), K' = 2 K, a_wrap = K gives A W_hi^T + A W_lo^T = A W^T
   * with W exact to ~22 bits -- the weights' fp16 rounding is a SYSTEMATIC perturbation of the model (it shifts the predicted camera
   * and the log-depth of every pixel coherently; tools/v1_precision_study.py), unlike the activations' rounding which averages out.
   * Three terms [A_hi | A_lo | A_hi] x [W_hi | W_hi | W_lo] (a_wrap = 2 K) give an fp32-class product. */
  int a_wrap, w_wrap;
  /* LayerNorm folded into a producer / consumer pair of GEMMs (large-tile kernel, dense A; reference metadinov2/block.py:85-89 norm1 /
   * norm2 between `x += proj(..)` / `x += fc2(..)` and the following qkv / fc1 Linear): no second pass over the fp32 stream.
   *   producer (UD_EPI_F32): row_stats_out [M][N / 64][2] fp32 = (sum, sum of squares) of the STORED fp32 row values over every
   *     64-column slab (one entry per wave of the tile; fixed order: bit-reproducible); out2 = the raw fp16 copy of the row.
   *   ud_row_stats_finalize: [M][slabs][2] -> [M][2] = (rstd, -mean * rstd), slabs <= 16 summed in one fixed butterfly order -- or
   *     row_stats_final below: the same reduction inside the producer, no extra launch.
   *   consumer (UD_EPI_F16 / UD_EPI_QKV): A = the raw fp16 copy, row_stats_in = the finalized [M][2]; the epilogue stores
   *     act(rstd * acc - mean * rstd * wsum[n] + bias[n]), wsum[n] = sum_k W[n, k] over the fp16-ROUNDED weights (the LayerNorm's affine
   *     folded into W / bias as before): rstd (x - mean) W^T = rstd (x W^T - mean 1 W^T).  fp16(x) carries the same 2^-11 relative
   *     rounding as fp16(LN(x)) when |mean| <~ std, which holds for residual streams (tap tests).
   *   ln_D / ln_eps: read by the producer when row_stats_final is set; ln_slabs: unused by the kernels (kept for program recording). */
  float* max_out;              /* UD_EPI_F32, optional: fp32 [*, ldc] laid out like `out`; max_init != 0: max_out = stored value, else max_out =
                                * max(max_out, stored value) -- the running element-wise maximum over the block outputs of a ConvNeXt stage
                                * (utils/misc.py:18-21 max_stack, unidepthv1/decoder.py:366-373) taken where the value is produced instead of by
                                * a separate pass over the stream (ud_max_f32: 36 launches, 12 B per element) */
  int max_init;
  int grp_rows;                /* internal (callers leave 0): set by ud_gemm_f16 when it runs a grouped problem -- `groups` GEMMs stacked along M --
                                * as ONE tile list of the large-tile kernel: rows per group */
  float* row_stats_out;
  float* row_stats_final;      /* producer, optional: [M][2] = (rstd, -mean * rstd) written by the producer ITSELF -- the last workgroup to finish a row
                                * tile (one ticket per row tile in row_stats_ticket: ceil(M / 128) unsigned, zeroed ONCE by the caller, never reset)
                                * reduces the partial sums of all column tiles in ascending slab order; replaces ud_row_stats_finalize (a ~7.5 us
                                * launch).  Large-tile kernel, tile-list schedule, N <= 1024; uses ln_D / ln_eps. */
  unsigned* row_stats_ticket;
  const float* row_stats_in;
  const float* wsum;
  int ln_slabs, ln_D;
  float ln_eps;
  /* UD_EPI_D2S, optional (round 4): the value the transposed convolution accumulates INTO is not read from `out` but interpolated on the fly
   * from up_src -- fp32 NHWC [*, up_img_rows, up_ld] holding a (up_H x up_W) map per image -- as its bilinear x2 up-sampling
   * (align_corners=False: exactly ud_upsample2x_nhwc mode 0, upsample.py:184-186 nn.Upsample behind the 1x1 conv), so the up-sampled map is
   * never written and read back: out = up2(up_src) + ConvT(A) (+ bias), out2 = act2(out).  Needs d2s_Hin * d2s_k == 2 * up_H (same for W). */
  const float* up_src;
  int up_H, up_W, up_ld, up_img_rows;
  /* Round 6: two-way K split of the LARGE-tile list (192 x 256 tiles): a one-round list of 32..128 tiles with K >= 2048 (dense A or zero-padded
   * 3x3 taps, fp16 / fp32 epilogues without row statistics; the decoder's stage-0 residual-conv-unit convolutions, layers/upsample.py:137-163)
   * runs as 2 * tiles workgroups, each over one half of K; same exchange, same splitk_cnt parity tickets (one unsigned per tile, zeroed once)
   * as the small-tile split above.  Scratch: splitk_ws of splitk_ws_bytes >= 2 * tiles * 192 * 256 * 4 bytes; a smaller (or absent)
   * scratch keeps the unsplit schedules.  0 for callers that only provide the small-tile scratch. */
  long long splitk_ws_bytes;
} UdGemm;

int ud_gemm_f16(const UdGemm* desc, void* stream);
/* kernel the call above would pick (profiling labels): 0/1/2 = 128-row tiles with BN 128/64/32, 3 = 192x256, 4 = 256x256, 5 = halo-tile conv,
 * 6 / 7 = 128x128 tiles, 4-stage pipelined ring without / with the two-way K split, 8 = row-balanced 256-column schedule;
 * + 16: folded-LayerNorm consumer instantiation, + 32: grouped problem run as one large-tile launch */
int ud_gemm_pick(const UdGemm* desc);

/* ---- LayerNorm (statistics only; the affine is folded into the consumer's weights at load time) ----
 * y(fp16)[orow, :] = (x[irow, :] - mean) * rsqrt(var + eps), biased variance (F.layer_norm).
 * Replaces nn.LayerNorm at metadinov2/block.py:85-89 (eps 1e-6, dinov2.py:167), dinov2.py:254,336-342 (eps 1e-5),
 * layers/attention.py:115-116, layers/mlp.py:29, decoder.py:186-188.
 * Row mapping: for r in [0, rows): img = r / rows_per_img, p = r % rows_per_img;
 *   irow = img * in_rows_per_img + p + in_row_off;  orow = img * out_rows_per_img + p + out_row_off. */
typedef struct UdLayerNorm {
  const float* x; void* y;
  int rows, D, ldx, ldy;
  float eps;
  int rows_per_img, in_rows_per_img, in_row_off, out_rows_per_img, out_row_off;
  int out_f32;            /* 0: y is fp16 (MFMA operand); 1: y is fp32 (the fp32 camera head) */
  const float* gamma;     /* optional affine [D] applied here (NULL = statistics only, the default: affines are folded into the consumer).  */
  const float* beta;      /* Needed where no linear consumer follows: the ConvNeXt stem's LayerNorm2d feeds a zero-padded depth-wise conv.  */
  const float* add;       /* optional fp32 [*, ldx] added to the input row BEFORE the statistics, indexed by the row's position inside its output
                           * image (out_row_off + p): `features + pos_embed` ahead of an MLP's norm (unidepthv1/decoder.py:80-83). */
  float* cls_y;           /* optional (round 6, fp16 output only): the launch covers rows_per_img + 1 rows per image (`rows` counts them); the FIRST of an */
  int ldcls;              /* image's rows -- input row in_row_off - 1, the class token in front of the patch tokens -- is normalised the same way and
                           * written as fp32 to cls_y[image * ldcls ..] (it feeds the fp32 camera head), the others as usual.  One launch instead of
                           * two per tapped layer of the encoder (dinov2.py:254 norm on x_norm_clstoken / x_norm_patchtokens). */
} UdLayerNorm;
int ud_layernorm_f32_f16(const UdLayerNorm* desc, void* stream);

/* partial (sum, sum of squares) pairs of UdGemm.row_stats_out [M][slabs][2] -> stats [M][2] = (rstd, -mean * rstd) with mean = S1 / D,
 * rstd = rsqrt(max(S2 / D - mean^2, 0) + eps) (biased variance, F.layer_norm): what UdGemm.row_stats_in consumes. */
int ud_row_stats_finalize(const float* partials, float* stats, int M, int slabs, int D, float eps, void* stream);

/* ---- fp32 small-M linear layer (camera head only) ---------------------------------------------------------
 * out[m, n] (+)= act(sum_k x[m,k] W[n,k] + bias[n] + add[m % add_mod, n]),  everything fp32, exact-erf GELU.
 * The camera head (decoder.py:48-114) maps 4 tokens per image to the pinhole parameters that generate EVERY ray; its
 * rounding error is amplified by the 2^k*pi ray-angle bands (decoder.py:246-252) into a common-mode depth error, so it is
 * the one place where fp16 MFMA operands are not accurate enough (measured: K error 1e-3 -> depth ARel 1e-3) and, at
 * M = 4*B rows, the one place where fp32 costs nothing.  Replaces nn.Linear in CameraHead / camera_token_adapter. */
typedef struct UdLinearF32 {
  const float* x; const float* W; const float* bias; const float* add; float* out;
  int M, N, K, ldx, ldw, ldc, ldadd, add_mod;
  int act, accumulate;
} UdLinearF32;
int ud_linear_f32(const UdLinearF32* desc, void* stream);

/* fp32 attention over T <= 8 tokens per image (camera head, layers/attention.py:81-165 with 4 tokens):
 * q [B*T, C], kv [B*T, 2C] = [K | V] heads-major, out [B*T, C]; H heads of width C/H. */
int ud_attention_small_f32(const float* q, const float* kv, float* out, int B, int T, int H, int C, float scale, void* stream);

/* ---- the whole camera head as ONE launch ------------------------------------------------------------------------
 * CameraHead.forward (decoder.py:94-108: project MLP, two AttentionBlocks over the 4 camera tokens of an image, out_pinhole MLP) and the four
 * camera_token_adapter Linears in front of it (decoder.py:34-45, 418-433) are ~26 dependent launches of the two fp32 kernels above, each a few
 * microseconds of work behind ~10 us of launch and fill latency (0.38 ms per infer() at bs = 8, 7 % of a bs = 1 call).  Here the chain is a
 * list of PHASES run by one persistent grid: phase = [LayerNorm of the input rows] -> out (+)= act(x W^T + bias + add) over a column slice per
 * workgroup, or the T-token attention per (image, head); workgroups meet at a grid barrier (one agent-scope counter) where a phase reads what
 * the previous one wrote.  Activations cross XCDs with system-scope (sc0 sc1) stores and loads -- the L2s of the 8 XCDs are not coherent with each
 * other inside a kernel -- while the next phase's weight slab streams into LDS under the barrier.  Same arithmetic as the launches it replaces
 * (fp32 FMA chains, two-pass LayerNorm, erf GELU); the summation order over k differs (32 interleaved slices per row, then a butterfly).
 * Limits (ud_camera_head_f32 returns UD_ERR_UNSUPPORTED otherwise and the caller keeps the per-layer launches): linear phases need K % 128 == 0,
 * ceil(N / workgroups) * K * 4 <= 32 KB, LayerNorm phases K <= 512; attention T <= 8, C / H <= 64; n_phases <= UD_CAM_MAX_PHASES. */
#define UD_CAM_MAX_PHASES 24
typedef struct UdCamPhase {
  const float* x;        /* linear: input rows [M, K] (ldx).  attention: packed rows [q | k | v] (ldx), q at column 0, k at C, v at 2 C */
  const float* W;        /* [N, K] row-major, dense (row stride K) */
  const float* bias;     /* [N] or NULL */
  const float* add;      /* [add_mod, ldadd] or NULL: out[m, n] += add[m % add_mod, n] for n < add_cols (latents_pos on the q columns) */
  float* out;            /* [M, N] (ldc) */
  int M, N, K, ldx, ldc, ldadd, add_mod, add_cols;
  int kind;              /* 0 linear, 1 attention */
  int ln;                /* 1: the input rows are layer-normalised first (statistics only: the affine is folded into W / bias by the caller) */
  int act, accumulate;   /* UD_ACT_NONE / UD_ACT_GELU; accumulate: out += ... */
  int sync;              /* 1: grid barrier after this phase (the next phase reads what this one wrote) */
} UdCamPhase;
typedef struct UdCameraHead {
  UdCamPhase ph[UD_CAM_MAX_PHASES];
  int n_phases;
  int T, H, C;           /* attention phases: tokens per image, heads, width (head width C / H) */
  float scale, eps;      /* softmax scale; LayerNorm eps */
  unsigned* sync_ws;     /* 16 words, zero before the FIRST launch; every launch leaves words 0 and 1 zero again.  Word 2 != 0 afterwards: a
                          * grid barrier timed out (the grid was not co-resident: CU mask, partitioned device, foreign kernels holding CUs).
                          * The failure is LOUD: every launch that ends with word 2 set overwrites the LAST phase's output (the camera
                          * parameters) with NaN, so intrinsics, rays and depth of that call are NaN; word 2 stays set (and keeps poisoning)
                          * until the caller clears it */
  int workgroups;        /* 0 = default (128) */
  unsigned* fail_host;   /* optional, HOST-mapped (pinned) word: set to 1 together with word 2 by a system-scope store, so the caller can poll
                          * the failure without a device synchronisation (unidepth_amd checks it at the start of the next infer()) */
  unsigned spin_limit;   /* polls of a barrier before a workgroup gives up; 0 = default (2^22: seconds).  Tests force the time-out with 1 */
} UdCameraHead;
/* Launches of ud_camera_head_f32 on one device are SERIALISED across streams (an event chain inside the library: the next launch waits for the
 * previous one wherever it was enqueued), so two spinning grids of one process are never resident together. */
int ud_camera_head_f32(const UdCameraHead* desc, void* stream);
int ud_camera_head_supported(const UdCameraHead* desc);   /* UD_OK if ud_camera_head_f32 would take this descriptor (host-side check, nothing is launched) */

/* ---- fused multi-head attention forward, head_dim 64 (padded), no mask, fp16 in/out, fp32 softmax ----
 * O = softmax(Q K^T * scale) V per (image, head).  Replaces F.scaled_dot_product_attention at
 * metadinov2/attention.py:58 and layers/attention.py:136-138 (and xformers memory_efficient_attention :77).
 * Q [img*q_rows_per_img + i, h*64 + d] (ldq), K likewise (ldk), Vt = V^T [img][h][64][kv_ld] in the UD_EPI_QKV block order
 * (zeros beyond Nk up to the next multiple of 64), O like Q (ldo).
 * kv_broadcast != 0: groups of kv_group consecutive images share one K/V image (single GT camera, decoder.py:400). */
typedef struct UdAttention {
  const void* Q; const void* K; const void* Vt; void* O;
  int B, H, Nq, Nk;
  int ldq, ldk, ldo, kv_ld;
  int q_rows_per_img, k_rows_per_img;
  float scale;
  int kv_broadcast;
  int kv_group;           /* with kv_broadcast: image i uses the K/V of image i / kv_group (0 -> all share image 0) */
  int q_prescaled;        /* != 0: Q already holds q * scale * log2(e) (folded into the q projection by the caller): `scale` is ignored and the
                           * kernel skips the per-score multiply (softmax(q k^T scale) is unchanged) */
} UdAttention;
int ud_attention_f16(const UdAttention* desc, void* stream);

/* ---- pre-processing + im2col for the 14x14 patch embedding ---------------------------------------------
 * Replaces unidepthv2.py:288-297 (/255, ImageNet mean/std, zero pad, bilinear align_corners=False resize) and
 * the unfold implied by Conv2d(k=s=14) (patch_embed.py:71-88).  rgb: uint8 (is_u8) or fp32 [B,3,H,W].
 * patches(fp16)[img*hw + py*w + px, c*196 + i*14 + j], row stride ldp (>= 588, pad cols untouched). */
typedef struct UdPreprocess {
  const void* rgb; void* patches;
  int B, H, W;            /* source image */
  int pad_l, pad_t, Hp, Wp;   /* padded size (before resize) */
  int Hn, Wn;             /* network input size (multiples of 14) */
  int ldp;
  int is_u8, normalize;
  float mean[3], inv_std[3];
} UdPreprocess;
int ud_preprocess_patches(const UdPreprocess* desc, void* stream);

/* ---- small helpers ----------------------------------------------------------------------------------- */
/* dst(fp32)[img*rows_per_img + row_off, :D] = src[:D]  (cls token + pos_embed[0]; dinov2.py:316-317) */
int ud_fill_rows_f32(float* dst, const float* src, int n_img, int rows_per_img, int row_off, int D, int ld, void* stream);

/* camera head tail (decoder.py:85-99 fill_intrinsics, :361-403 run_camera; unidepthv2.py:92-108 _postprocess_intrinsics):
 * raw[(b*4 + j) * raw_stride] = j-th raw camera parameter of image b -> intr4 [B,4] = (fx,fy,cx,cy) at network
 * resolution, K33 [B,9] row-major, Kinv33 [B,9] (closed-form pinhole inverse), Kpost33 [B,9] = K with
 * fx,fy,cx,cy /= resize_factor and cx -= pad_l, cy -= pad_t (the matrix infer() returns). */
int ud_camera_intrinsics(const float* raw, int raw_stride, float* intr4, float* K33, float* Kinv33, float* Kpost33,
                         int B, int Hn, int Wn, float resize_factor, int pad_l, int pad_t, void* stream);

/* rays [nb,3,Hn,Wn] fp32 = normalise(Kinv @ [u+0.5, v+0.5, 1]) (utils/coordinate.py:4-20 pixel centres).
 * gt_mode 0: predicted camera, norm clamp 1e-5 (decoder.py:389-393);
 * gt_mode 1: user camera (utils/camera.py:254-266 Pinhole.unproject: divide by z.clip(1e-4); :88-92 norm clamp 1e-4).
 * gt_mode 2 / 3: user EUCM / Spherical camera (utils/camera.py:307-328 / :371-386 unproject + :88-92 get_rays); the 9 floats per
 *   camera are then the model parameters at network resolution: (fx, fy, cx, cy, alpha, beta, -, -, -) resp.
 *   (fx, fy, cx, cy, width, height, hfov/2, vfov/2, -) instead of an inverse intrinsic matrix. */
int ud_rays_from_kinv(const float* Kinv33, float* rays, int nb, int Hn, int Wn, int gt_mode, void* stream);

/* rays [1,3,Hn,Wn] fp32 of ONE user camera whose unprojection is iterative (utils/camera.py get_rays :88-92 over
 * OPENCV.unproject :496-694 (model 4), Fisheye624.unproject :778-974 (model 5), MEI.unproject :985-1082 (model 6)).
 * params: device pointer, 16 floats [fx, fy, cx, cy, k1..k6, p1, p2, s1..s4] (models 4, 5; OPENCV: k4..k6 = 0) or 9 floats
 * [fx, fy, cx, cy, k1, k2, p1, p2, xi] (model 6), at network resolution.  Which distortion groups are active is decided on
 * the device from the parameters, like the reference's use_radial / use_tangential / use_thin_prism.
 * scratch: 4*Hn*Wn + 16 floats (models 4, 5; may be NULL for MEI) -- per-pixel solver state plus the per-iteration maximum
 * residual that reproduces the reference's image-wide early exit of the radial trust-region loop (:629-631). */
int ud_rays_from_camera(const float* params, float* rays, float* scratch, int Hn, int Wn, int model, void* stream);

/* ray embedding (decoder.py:234-253): antialiased bilinear down-sample of rays [nb,3,Hn,Wn] to (h,w)
 * (utils/geometric.py:227-252), renormalise (clip 1e-4), polar/azimuth, C/2 log-spaced sine bands each
 * (utils/positional_embedding.py:218-256; `scales` = the C/2 band frequencies, fp32), then LayerNorm statistics
 * (eps) -> xhat fp16 [img*rows_per_img + token, ldy]. */
typedef struct UdRayEmbed {
  const float* rays; const float* scales; void* xhat;
  int nb, Hn, Wn, h, w, C, ldy, rows_per_img;
  float eps;
} UdRayEmbed;
int ud_ray_embed(const UdRayEmbed* desc, void* stream);

/* x2 bilinear up-sampling, align_corners=False (nn.Upsample in layers/upsample.py:215-217), NHWC fp32 in [B,H,W,C] (row stride ldin).
 * mode 0: out fp32 [B,2H,2W,C]; mode 1: out = LayerNorm-statistics(fp16) of the up-sampled pixel (eps), row stride ldy. */
typedef struct UdUpsample2x {
  const void* in; void* out;
  int B, H, W, C, ldin, ldy, mode;
  float eps;
  int in_img_rows;   /* rows (pixels) between consecutive images of `in`; 0 -> H*W */
} UdUpsample2x;
int ud_upsample2x_nhwc(const UdUpsample2x* desc, void* stream);

/* bilinear resize, align_corners=True (decoder.py:299-301,309-311), NHWC fp16 -> NHWC fp16, G groups */
typedef struct UdResizeAC {
  const void* in; void* out;
  int G, B, Hin, Win, Hout, Wout, C;
} UdResizeAC;
int ud_resize_ac_nhwc_f16(const UdResizeAC* desc, void* stream);

/* output assembly (decoder.py:456-462, unidepthv2.py:375-377, :310-339): fp32 NCHW resize (F.interpolate align_corners=False,
 * bilinear or bicubic = the two modes the reference's `interpolation_mode` accepts) + crop of _postprocess (unidepthv2.py:80-89). */
typedef struct UdFinalize {
  const float* radius_net;  /* [B,Hn,Wn] */
  const float* conf_net;    /* [B,Hn,Wn] */
  const float* rays_net;    /* [nb_rays,3,Hn,Wn] */
  float* confidence; float* radius; float* depth; float* points; float* rays;   /* outputs at [.,.,Ho,Wo] */
  int B, nb_rays, Hn, Wn;
  int Hp, Wp;               /* padded (pre-crop) size the network maps are resized to */
  int pad_l, pad_t, Ho, Wo; /* crop */
  int mode;                 /* resampling of _postprocess (unidepthv2.py:80-89, `interpolation_mode`): 0 bilinear, 1 bicubic; align_corners=False */
} UdFinalize;
int ud_finalize_outputs(const UdFinalize* desc, void* stream);

/* NHWC fp32 (row stride ld, rows_per_img rows per image) -> NCHW fp32 [B,C,h*w]  (depth_features, decoder.py:265-267) */
int ud_nhwc_to_nchw_f32(const float* in, float* out, int B, int hw, int C, int ld, int rows_per_img, void* stream);

/* ---- ConvNeXt-side ops of the UniDepthV1 path (reference backbones/convnext.py; decoder blocks layers/convnext.py) ---------------
 * Depth-wise 7x7 convolution, zero padding 3 (convnext.py:171-179 conv_dw via timm create_conv2d(depthwise=True); layers/convnext.py:16-24):
 * y[b,y,x,c] = bias[c] + sum_{ky,kx} w[(ky*7+kx)*C + c] * x[b, y+ky-3, x+kx-3, c];  NHWC fp32 in / out, pixel strides ldx / ldy (floats);
 * w is TAP-major [49][C] (repacked from the reference's [C,1,7,7] at load time). */
typedef struct UdDwConv7 {
  const float* x; const float* w; const float* bias; float* y;
  int B, H, W, C, ldx, ldy;
  /* Round 6 (VERDICT r5 missing #5c), the LayerNorm behind the convolution (convnext.py:215-216 `x = self.norm(x)`) folded into the producer / consumer pair:
   * y16 (optional; then y may be NULL): the convolution's output as RAW fp16 [pixels, ldy16] -- the A operand of a LayerNorm-folded consumer GEMM
   * (UdGemm.row_stats_in); stats_out (with y16): per pixel and 64-channel slab the (sum, sum of squares) of the fp32 outputs, [pixels][C / 64][2] =
   * UdGemm.row_stats_out's layout, reduced by ud_row_stats_finalize (C / 64 <= 16).  C % 64 == 0 (the LDS-tiled kernel). */
  void* y16; float* stats_out; int ldy16;
  /* optional, with stats_out: the LAST of a pixel tile's C / 64 channel blocks to finish (one ticket per 8 x 16 tile of output pixels: stats_ticket, >= number of
   * tiles unsigned words, zeroed once by the caller, self-resetting) reduces the tile's partial sums in slab order and writes stats_final [pixels][2] =
   * (rstd, -mean * rstd) with eps ln_eps -- what ud_row_stats_finalize would write; no reduction launch. */
  float* stats_final; unsigned* stats_ticket; float ln_eps;
} UdDwConv7;
int ud_dwconv7_nhwc_f32(const UdDwConv7* desc, void* stream);
/* LayerNorm2d statistics (eps; affine folded into the conv weights) of every pixel of x fp32 NHWC [B,H,W,C], written as fp16 into the
 * im2col image of the following Conv2d(k=2, s=2, pad 0) (convnext.py:245-266 ConvNeXtStage.downsample): row (b, y/2, x/2), columns
 * ((y&1)*2 + (x&1))*C + c, row stride ldo >= 4C; an odd last row / column of x is dropped like the convolution drops it. */
int ud_layernorm_patchify2(const float* x, void* out, int B, int H, int W, int C, int ldo, float eps, void* stream);
/* im2col of the 4x4 stride-4 stem convolution (convnext.py:370-383): img fp32 NCHW [B,3,H,W] -> fp16 [B*(H/4)*(W/4), ldo], column
 * c*16 + ky*4 + kx (48 used; the caller zero-fills the pad columns once). */
int ud_patchify4_nchw(const float* img, void* out, int B, int H, int W, int ldo, void* stream);
/* dst = init ? src : max(dst, src), element-wise over n fp32 values (utils/misc.py:18-21 max_stack over a stage's block outputs). */
int ud_max_f32(float* dst, const float* src, long long n, int init, void* stream);
/* out[b*ldo + c] = mean over the HW pixels of x[b, :, c] (x fp32 [B,HW,C]): the "class tokens" of the ConvNeXt wrapper (convnext.py:458). */
int ud_spatial_mean_f32(const float* x, float* out, int B, int HW, int C, int ldo, void* stream);

/* ---- decoder-side ops of the UniDepthV1 path that are not GEMMs / attention / LayerNorm: one entry point, dispatched on `kind` ----
 * a, b, c: inputs; out, out2: outputs (c is an output for UD_V1_CAMERA); i[], f[]: per-kind integers / floats as listed.
 *  RESIZE_AA     F.interpolate(bilinear, align_corners=False, antialias=True) of a crop window of an NHWC fp32 map (utils/geometric.py:227-252
 *                flat_interpolate; unidepthv1.py:66-86 _postprocess).  i = B, Hi, Wi, Ho, Wo, C, ldi, ldo, y0, x0, Hc, Wc (C % 4 == 0)
 *  SH_EMBED      rays [nb,3,Hn,Wn] -> antialiased resize to (h, w), F.normalize, 81 real spherical harmonics (utils/sht.py:833 rsh_cart_8),
 *                LayerNorm statistics (eps f[0]) -> fp16 [nb*rows_per_img, ldo >= 128] (decoder.py:205-222).  i = nb, Hn, Wn, h, w, ldo, rows_per_img
 *  SOFTMAX       out[r, :N] = softmax(f[0] * a[r, :N]) rows of fp32 scores -> fp16 (or fp32), pad columns N..ldo zero (the softmax inside
 *                F.scaled_dot_product_attention for single-head width-512 attention, layers/attention.py:136).
 *                i = rows & 0x7fffffff, N, ldi, ldo, out_f32, rows >> 31
 *  ATTN_FEWQ     T <= 8 queries against Nk keys, ONE head of width D (camera head `aggregate`, decoder.py:94, layers/attention.py:81-165):
 *                a = q fp32 [B*T, D], b = kv fp16 [B*Nk, 2D] = [K | V], out fp32 [B*T, D].  i = B, T, Nk, D; f[0] = scale
 *                c (optional) = fp32 scratch of B * ceil(Nk / 64) * T * (D + 2) floats: the keys are then processed in chunks of 64 by
 *                separate workgroups and merged in a second launch (deterministic order)
 *  HEAD_MIX      the attention of layers_8 / layers_4 AS THE REFERENCE COMPUTES IT (layers/nystrom_attention.py:59-62,81: q, k, v go to xformers
 *                NystromAttention as [b, n, h, d]; its `seq_len = k.size(-2)` is then h <= num_landmarks, so it takes its plain-softmax branch
 *                over the last two axes): per token, softmax((q_i / sqrt d) . k_j over the h heads j) v_j.  a = q fp32 [M, ldq], b = [K | V] fp32
 *                [M, ldkv], out fp16 [M, ldo]; head width 64.  i = M, h (2 / 4 / 8), ldq, ldkv, ldo; f[0] = 1 / sqrt(d)
 *  ADD           out = a + b (fp32; latents + ray embedding, decoder.py:263,283,303).  i = n & 0x7fffffff, n >> 31
 *  COPY_ROWS     out[(img*rows_per_img + row_off + t)*ld + d] = a[(img*T + t)*D + d] (torch.cat of token groups).  i = n_img, T, rows_per_img, row_off, D, ld, to_f16
 *                to_f16 == 2 (ld >= 2 D): the value as TWO fp16 terms, hi at column d and lo = fp16(x - hi) at column D + d -- the A operand
 *                [A_hi | A_lo] of a three-term product against [W_hi | W_hi | W_lo] (UdGemm.a_wrap = 2 K)
 *  RESIZE_AC_SPLIT  nn.UpsamplingBilinear2d (align_corners=True, layers/upsample.py:34) of an fp32 NHWC map, written as [hi | lo] fp16
 *                (2 C channels per pixel) for the 3x3 convolution behind it.  i = B, Hin, Win, Hout, Wout, C (C % 4 == 0)
 *  CAMERA        raw [B*4] -> K33 (out), its inverse (out2), post-processed K (c) (decoder.py:85-99,347-353; unidepthv1.py:88-92).
 *                i = B, Hn, Wn, pad_l, pad_t; f[0] = ratio
 *  POINTS        z map + K33 -> points [B,3,H,W] (out), depth [B,1,H,W] (out2) (unidepthv1.py:353-371; utils/geometric.py:45-73).  i = B, H, W, ldz, nK
 *  MEAN3         out = (a + b + c) / 3 on column 0 of strided maps (unidepthv1.py:66-77).  i = n & 0x7fffffff, ld, n >> 31
 *  VIT_TAP       UniDepthV1 on a DINOv2 backbone: out = init ? v : max(out, v), v = patch tokens + class token of the block whose residual
 *                stream is a [B*Np, D] (row 0 of an image = class token): unidepthv1.py:324-328 + decoder.py:366-373 max_stack; out2
 *                (optional) = the raw class tokens [B, D].  i = B, Np, hw, D, init
 *  OUT_CONV3     out[pixel, 0] = exp(clamp(conv3x3(x)[pixel] + f[0], -10, 10)) for ONE output channel, zero padding, everything fp32: the multi-scale
 *                outputs out8 / out4 / out2 (unidepthv1/decoder.py:185-187 nn.Conv2d(C, 1, 3, padding = 1), :250-298).  a = x NHWC [B, H, W, C],
 *                b = w [9, C] (tap = ky * 3 + kx), out [B * H * W, ldo].  i = B, H, W, C (% 4 == 0), ldo
 *  PREPROCESS    V1 network image (unidepthv1.py:305-321,50-56): [/255], ImageNet normalise, antialiased resize to (h, w), zero pad to (Hn, Wn).
 *                i = B, H, W, h, w, Hn, Wn, pad_l, pad_t, is_u8, div255, normalize */
enum { UD_V1_RESIZE_AA = 1, UD_V1_SH_EMBED = 2, UD_V1_SOFTMAX = 3, UD_V1_ATTN_FEWQ = 4, UD_V1_HEAD_MIX = 5, UD_V1_ADD = 8, UD_V1_COPY_ROWS = 9,
       UD_V1_CAMERA = 11, UD_V1_POINTS = 12, UD_V1_MEAN3 = 13, UD_V1_PREPROCESS = 14, UD_V1_VIT_TAP = 15, UD_V1_RESIZE_AC_SPLIT = 17, UD_V1_OUT_CONV3 = 18 };
typedef struct UdV1Op {
  int kind;
  const void* a; const void* b; void* c; void* out; void* out2;
  int i[12];
  float f[4];
} UdV1Op;
int ud_v1_op(const UdV1Op* desc, void* stream);

/* ---- the exchange step of batch data parallelism: all-gather of the packed output rows over RCCL / xGMI ----
 * The reference has no distributed inference path (unidepth/utils/distributed.py:153-176 sync_tensor_across_gpus is its pad -> gather -> trim
 * helper for variable-length gathers; unidepth_amd/dist.py keeps that pattern on the host).  One process per GPU, one communicator per
 * process; librccl is opened on first use.  Rank 0 calls ud_rccl_unique_id and hands the 128 bytes to the other ranks by any host channel;
 * every rank then calls ud_rccl_init with the calling thread's HIP device set.
 *   ud_rccl_allgather_outputs: recv [world][bytes_per_rank] <- every rank's `send` [bytes_per_rank] (device buffers, equal size on all ranks),
 *   enqueued on `stream`; direct != 0: the all-pairs send / receive group (every peer over its own xGMI link) instead of ncclAllGather. */
int ud_rccl_unique_id(void* id128);
int ud_rccl_init(const void* id128, int world, int rank);
int ud_rccl_allgather_outputs(const void* send, void* recv, size_t bytes_per_rank, int direct, void* stream);
int ud_rccl_finalize(void);

/* ---- the reference's two native extensions (evaluation / loss side), forward only ----
 * K nearest neighbours of every p1 point among the p2 points of the same cloud: replaces KNN.knn_points_idx
 * (unidepth/ops/knn/src/knn_ext.cpp:8 -> knn.h:44-80 KNearestNeighborIdx -> knn.cu:314-419 KNearestNeighborIdxCuda; caller
 * functions/knn.py:72 _knn_points.forward <- utils/chamfer_distance.py:143-144 <- utils/evaluation_depth.py:12-34).
 *   p1 [N,P1,D], p2 [N,P2,D] fp32 contiguous; lengths1/lengths2 int64 [N] or NULL (= P1 / P2 everywhere).
 *   dists fp32 [N,P1,K] squared L2 (norm 2) or L1 (norm 1) distances, idx int64 [N,P1,K]; every element is written (padding = 0).
 *   The K neighbours are the K smallest (dist, index) pairs in ascending order, i.e. already what functions/knn.py:75-91 obtains by
 *   sorting; ties resolve to the lower index (knn_cpu.cpp:40-58).  1 <= D <= 32 (D <= 8 on the register path), 1 <= K <= 32.
 *   work: optional u64 [N*P1] scratch; when given and K == 1, small P1 problems are split over P2 (ud_knn_split slices). */
typedef struct UdKnn {
  const float* p1; const float* p2;
  const long long* lengths1; const long long* lengths2;
  float* dists; long long* idx;
  unsigned long long* work;
  int N, P1, P2, D, K, norm;
} UdKnn;
int ud_knn_points(const UdKnn* desc, void* stream);
int ud_knn_split(const UdKnn* desc);
/* Patch gather: replaces RandomPatchExtraction.extract_patches_forward (unidepth/ops/extract_patches/src/extract_patches.cpp:3-6 ->
 * src/cuda/extract_patches_kernel.cu:9-35,65-95) together with the zero padding its module does first (modules/patch_extractor.py:26-42).
 *   in fp32 [B,C,H,W]; centers int32 [B,N,2] = (y, x) in the coordinates of the image padded by (pad_h, pad_w) on every side (pass
 *   pad = 0 for raw coordinates); out fp32, B*N*C*h*w elements in [b][n][c][i][j] order -- the order the reference kernel writes
 *   (extract_patches_kernel.cu:91), which it then views as {B,C,N,h,w} (:22).  Pixels outside the image read as 0. */
typedef struct UdExtractPatches {
  const float* in; float* out; const int* centers;
  int B, C, H, W, N, h, w, pad_h, pad_w;
} UdExtractPatches;
int ud_extract_patches(const UdExtractPatches* desc, void* stream);

/* ---- launch programs: a recorded list of the ops above replayed with one call (host-side runtime) ---- */
typedef struct UdProgram UdProgram;
UdProgram* ud_program_create(void);
void ud_program_destroy(UdProgram*);
int ud_program_size(const UdProgram*);
int ud_program_add_gemm(UdProgram*, const UdGemm*);
int ud_program_add_layernorm(UdProgram*, const UdLayerNorm*);
int ud_program_add_row_stats_finalize(UdProgram*, const float* partials, float* stats, int M, int slabs, int D, float eps);
int ud_program_add_attention(UdProgram*, const UdAttention*);
int ud_program_add_linear_f32(UdProgram*, const UdLinearF32*);
int ud_program_add_camera_head(UdProgram*, const UdCameraHead*);
int ud_program_add_attention_small_f32(UdProgram*, const float* q, const float* kv, float* out, int B, int T, int H, int C, float scale);
int ud_program_add_preprocess(UdProgram*, const UdPreprocess*);
int ud_program_add_fill_rows(UdProgram*, float* dst, const float* src, int n_img, int rows_per_img, int row_off, int D, int ld);
int ud_program_add_camera_intrinsics(UdProgram*, const float* raw, int raw_stride, float* intr4, float* K33, float* Kinv33, float* Kpost33,
                                     int B, int Hn, int Wn, float resize_factor, int pad_l, int pad_t);
int ud_program_add_rays(UdProgram*, const float* Kinv33, float* rays, int nb, int Hn, int Wn, int gt_mode);
int ud_program_add_rays_camera(UdProgram*, const float* params, float* rays, float* scratch, int Hn, int Wn, int model);
int ud_program_add_ray_embed(UdProgram*, const UdRayEmbed*);
int ud_program_add_upsample2x(UdProgram*, const UdUpsample2x*);
int ud_program_add_resize_ac(UdProgram*, const UdResizeAC*);
int ud_program_add_finalize(UdProgram*, const UdFinalize*);
int ud_program_add_nhwc_to_nchw(UdProgram*, const float* in, float* out, int B, int hw, int C, int ld, int rows_per_img);
int ud_program_add_dwconv7(UdProgram*, const UdDwConv7*);
int ud_program_add_layernorm_patchify2(UdProgram*, const float* x, void* out, int B, int H, int W, int C, int ldo, float eps);
int ud_program_add_patchify4(UdProgram*, const float* img, void* out, int B, int H, int W, int ldo);
int ud_program_add_max(UdProgram*, float* dst, const float* src, long long n, int init);
int ud_program_add_spatial_mean(UdProgram*, const float* x, float* out, int B, int HW, int C, int ldo);
int ud_program_add_v1_op(UdProgram*, const UdV1Op*);
/* run ops [first, last) on `stream`; returns 0 or the first failing op's error code.  Stateless: a recorded program is never modified by a replay */
int ud_program_run(const UdProgram*, int first, int last, void* stream);

/* ---- measurement support (bench.py `roofline.attainable_this_box`; not on any inference path): one launch of a pure MFMA instruction stream
 * (v_mfma_f32_32x32x16_f16, 16 independent instructions per iteration and wave, 4 waves per workgroup) on the random fp16 values in `operands`
 * (2 MiB); `sink`: workgroups * 256 floats, never written for finite operands; *flop_out = FLOP of the launch.  What the matrix pipes of this box
 * sustain at their power-limited clock: boxes of one pool differ by several per cent, the datasheet peak (2.5 PFLOP/s) is a constant. */
int ud_calib_mfma_stream(const void* operands, int iters, int workgroups, void* sink, double* flop_out, void* stream);
/* the same for v_mfma_f32_16x16x32_f16 (the GEMM family's instruction): workgroups of 8 waves, sink >= workgroups * 512 floats (round 6) */
int ud_calib_mfma_stream16(const void* operands, int iters, int workgroups, void* sink, double* flop_out, void* stream);

/* library info; ud_struct_size(i): sizeof the i-th descriptor struct in declaration order (UdGemm = 0 ... UdLinearF32 = 8, UdDwConv7 = 9, UdV1Op = 10, UdKnn = 11, UdExtractPatches = 12, UdCameraHead = 13) */
int ud_version(void);
int ud_struct_size(int which);
const char* ud_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
