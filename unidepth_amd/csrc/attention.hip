// Fused multi-head attention forward for gfx950 (flash style, online softmax), head_dim = 64, fp16 operands,
// fp32 scores / statistics / accumulators.  Replaces F.scaled_dot_product_attention at metadinov2/attention.py:51-62 (encoder) and
// layers/attention.py:136 (decoder).  One workgroup = NW waves x 32 query rows; K and V^T tiles of 64 keys go
// HBM/L2 -> LDS by buffer_load ... lds (no VGPR staging, no ds_write pass), double buffered.
//
// MFMA orientation (v_mfma_f32_32x32x16_f16), chosen so that softmax is lane-local:
//   S^T[key][q] = K[key][:] . Q[q][:]      (A = K fragment from LDS, B = Q fragment held in registers)
//     -> lane (q = lane & 31) holds 16 of the 32 keys of a block, the partner lane ^ 32 the other 16:
//        row max / row sum = in-register reduction + one cross-lane exchange.
//   O^T[d][q]  += V^T[d][key] . P^T[key][q] (A = V^T fragment from LDS, B = the exponentiated S^T registers,
//        re-used in place: the accumulator layout of S^T is exactly a valid B-operand layout once the k-slots
//        of the MFMA are mapped to keys {4h+8g+e}; the V^T fragment is read with the same key permutation).
//     -> the per-row rescale factor alpha is per lane, no shuffles in the main loop.
// V arrives pre-transposed AND in that k-slot order ([head][d][key'], 4-key blocks of every 16 keys stored [0, 2, 1, 3];
// written that way by the QKV GEMM epilogue), so the 8 keys a lane feeds to one MFMA are 16 contiguous bytes of a row and
// both tiles are plain row-contiguous copies.  LDS layouts: K [64 keys][64 d] and V^T [64 d][64 keys] halves, 128-byte
// rows, the 16-byte chunk index XOR-swizzled by (row >> 1) & 7 -- applied to the per-lane global SOURCE address of the
// DMA (its LDS destination is lane-linear) and to the ds_read_b128 address: conflict-free fragment reads.
//
// Two kernels:
//   attention_pipe_kernel  (round 5) the product path for pre-scaled Q (UdAttention.q_prescaled: encoder and decoder): the tile loop is
//       SOFTWARE-PIPELINED inside every wave -- while the matrix pipe works on Q K^T of tile t+1 and P V of tile t, the same wave's
//       VALU turns the scores of tile t into probabilities and reduces the row maximum of tile t+1, so MFMA and VALU instructions of
//       INDEPENDENT data alternate in one instruction stream (the round-4 counters showed the one-tile-at-a-time form running the two
//       pipes one after the other: MFMA busy 37 % + VALU busy 59 % = 96 % of the SIMD cycles, with four waves per SIMD).
//   attention_kernel       the one-tile-at-a-time form (rounds 1-4): raw Q (MODE 0: one fma per score), and the tested fallback of the
//       pipelined kernel (-DUD_ATTN_PIPE=0).
#include "ud_common.h"
#include <type_traits>

#ifndef UD_ATTN_PIPE
#define UD_ATTN_PIPE 1          // 0: pre-scaled Q takes the one-tile-at-a-time kernel (MODE 1) like rounds 2-4
#endif
// Fixed choices of the pipelined kernel, each measured on the encoder shape (B = 8, H = 16, N = 1370; profiles/r05_attn_*.txt) -- the alternatives
// were built as compile-time variants for the A/B runs and removed afterwards:
//   OPT 3 (offset as MFMA C operand + matrix-pipe row sums) 91.1 us | without the C operand 93.2 | without the row-sum MFMAs 91.0 | neither 92.5
//   4 waves per workgroup 91.1 | 8 waves 96.6 | 2 waves 95.3;  fragment ring 3 deep 89.5 | 4 deep 90.5 (5 spilled registers outside the loop)
//   two tiles per loop trip (every LDS address = base + immediate, 252 registers, 2 waves per SIMD) 93.5 | one tile per trip (149-173 registers:
//   3 waves per SIMD, but 33 more VALU moves per tile) 92.6-98.0
#define UD_ATTN_PIPE_OPT 3
#define UD_ATTN_PIPE_MINW 2
#define UD_ATTN_PIPE_NW 4
#define UD_ATTN_PIPE_FD 3
#define UD_ATTN_PIPE_WGS_PER_XCD 64      // persistent workgroups per XCD (two 256-thread workgroups on each of its 32 CUs); measured against one workgroup
                                         // per item: 89.5 vs 91.2 us alone, 615.6 vs 613.0 images/s in the step (profiles/r05_attn_persistent_ab.txt)

namespace {

constexpr int KT = 64;             // keys per tile
constexpr int KS_BYTES = 64 * 128;
constexpr int VS_BYTES = 64 * 128;
constexpr int STAGE = KS_BYTES + VS_BYTES;

// XCD-aware work map shared by both kernels (1-D grid; workgroup b runs on XCD b % 8): the q-tiles of one (image, head) pair share its
// K / V^T (2 x 175 KB at N = 1370), so all of them go to ONE XCD, in consecutive dispatch slots -- with the natural 3-D grid they
// were spread over all 8 private L2s and every XCD pulled nearly every K/V through the fabric (rocprofv3 FETCH_SIZE
// 403 MB per launch against 67 MB of unique Q/K/V: the kernel ran at the fabric read rate, 4.1 TB/s, not at MFMA rate).

// MODE 0: Q as given, P = exp2(S c - m c) with c = scale log2(e) (one fma per score).
// MODE 1: Q arrives PRE-SCALED by c (UdAttention.q_prescaled: the engine folds c into the q projection's weights at load time), and
//   the running maximum is subtracted by the MFMA itself -- the score accumulators start at -m instead of 0 -- so P = exp2(S'') with
//   no VALU in between.  A growing maximum (deferred, threshold 2^8) is handled on the rare path by shifting S''.
// NW = waves per workgroup, 32 query rows each.
template <int MODE, int NW = 4>
__global__ __launch_bounds__(NW * 64) void attention_kernel(const UdAttention p, const float defer_thr) {
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5;
  const int ql = lane & 31;
  const int qt = (p.Nq + NW * 32 - 1) / (NW * 32);
  const int pairs = p.B * p.H;
  const int nt = (p.Nk + KT - 1) / KT;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int pr = xcd + 8 * (slot / qt);                   // all q-tiles of a pair on one XCD, consecutive slots
  if (pr >= pairs) return;
  const int head = pr % p.H;
  const int img = pr / p.H;
  const int kimg = p.kv_broadcast ? (p.kv_group > 0 ? img / p.kv_group : 0) : img;
  const int q0 = (slot % qt) * (NW * 32) + wv * 32;
  const int kt0 = 0, kt1 = nt;

  const half_t* Q = (const half_t*)p.Q;
  const half_t* K = (const half_t*)p.K;
  const half_t* Vt = (const half_t*)p.Vt + ((size_t)kimg * p.H + head) * 64 * (size_t)p.kv_ld;

  // ---- Q fragments (B operand): q = ql, d = ks*16 + hh*8 .. +8
  half8 qf[4];
  {
    int qr = q0 + ql;
    qr = qr < p.Nq ? qr : p.Nq - 1;
    const half_t* qp = Q + ((size_t)img * p.q_rows_per_img + qr) * p.ldq + head * 64 + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const half8*)(qp + ks * 16);
  }

  // ---- tile loader: 512 16-byte chunks per operand tile = 8 wave-instructions of 8 rows x 128 B; wave w issues pieces w*PW .. +PW.
  // Buffer-descriptor DMA: V^T advances by an SGPR offset (no VALU); K by one v_add per piece, and its descriptor ends after
  // key Nk-1, so the rows of the last tile beyond the sequence read as zeros instead of the next image's keys.
  const int lrow = lane >> 3, lch = lane & 7;
  const ud_rsrc_t rK = ud_make_rsrc(K + (size_t)kimg * p.k_rows_per_img * p.ldk + head * 64, (unsigned)((p.Nk - 1) * p.ldk + 64) * 2u);
  const ud_rsrc_t rV = ud_make_rsrc(Vt, 64u * (unsigned)p.kv_ld * 2u);
  constexpr int PW = 8 / NW;                              // 1-KiB pieces per wave and operand tile (8 pieces each)
  unsigned koff[PW], voff[PW];
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int row = (wv * PW + i) * 8 + lrow;             // key (K tile) / d (V^T tile)
    const int ch = lch ^ ((row >> 1) & 7);
    koff[i] = (unsigned)(row * p.ldk + ch * 8) * 2u;
    voff[i] = (unsigned)(row * p.kv_ld + ch * 8) * 2u;
  }
  const unsigned kstep = (unsigned)(KT * p.ldk) * 2u;
  auto issue = [&](int kt, int stage) {
    char* sb = smem + stage * STAGE;
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const int piece = wv * PW + i;
      ud_bufl16(rK, koff[i] + (unsigned)kt * kstep, 0, sb + piece * 1024);
      ud_bufl16(rV, voff[i], kt * (KT * 2), sb + KS_BYTES + piece * 1024);
    }
  };

  f32x16 o[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.0f;
  float m_i = MODE ? 0.0f : -1.0e30f, l_i = 0.0f;
  const float c = p.scale * 1.4426950408889634f;

  if (kt0 < kt1) issue(kt0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int kswz = (ql >> 1) & 7;   // (key >> 1) & 7 for key = kb*32 + ql
  int st_cur = 0;
  for (int kt = kt0; kt < kt1; ++kt) {
    if (kt + 1 < kt1) issue(kt + 1, st_cur ^ 1);        // every wave passed the barrier that ended tile kt-1, the last reader of that slot
    const char* sb = smem + st_cur * STAGE;
    // ---- S^T = K Q^T  (two 32-key blocks), MODE 1: minus the running offset m_i
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = MODE ? -m_i : 0.0f;
      const char* kp = sb + (kb * 32 + ql) * 128;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const half8 kf = *(const half8*)(kp + (((ks * 2 + hh) ^ kswz) << 4));
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[kb], 0, 0, 0);
      }
    }
    // ---- mask the key tail (last tile only)
    if (kt == nt - 1 && (p.Nk & (KT - 1))) {
      const int kbase = kt * KT + 4 * hh;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase + kb * 32 + (r & 3) + 8 * (r >> 2);
          if (key >= p.Nk) s[kb][r] = -1.0e30f;
        }
    }
    // ---- online softmax (lane-local + partner lane ^ 32), max update deferred until it grows by > 2^8 (fp16 P has
    //      constant relative precision, so P up to 256 costs no accuracy; accumulation is fp32)
    float mt = s[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) mt = fmaxf(fmaxf(mt, s[kb][r]), s[kb][r + 1]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    float ls = 0.0f;
    half8 pf[2][2];
    if constexpr (MODE == 0) {
      if (__any((mt - m_i) * c > defer_thr)) {
        const float m_new = fmaxf(m_i, mt);
        const float alpha = __builtin_amdgcn_exp2f((m_i - m_new) * c);
        m_i = m_new;
        l_i *= alpha;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
      }
      const float mc = m_i * c;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            f32x2 pv;
            pv[0] = __builtin_amdgcn_exp2f(fmaf(s[kb][t * 8 + e], c, -mc));
            pv[1] = __builtin_amdgcn_exp2f(fmaf(s[kb][t * 8 + e + 1], c, -mc));
            ls += pv[0] + pv[1];
            const half2v ph = __builtin_convertvector(pv, half2v);      // v_cvt_pk_f16_f32 (round to nearest even)
            pf[kb][t][e] = ph[0];
            pf[kb][t][e + 1] = ph[1];
          }
    } else {
      // scores are already S - m (in log2 units); mt = how far this tile's row maximum exceeds the running one
      if (kt == kt0 || __any(mt > defer_thr)) {
        const float d = kt == kt0 ? mt : fmaxf(mt, 0.0f);
        m_i += d;
        if (kt != kt0) {                                   // first tile: O and l are still zero (and exp2(-d) may overflow)
          const float alpha = __builtin_amdgcn_exp2f(-d);
          l_i *= alpha;
#pragma unroll
          for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[kb][r] -= d;
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            f32x2 pv;
            pv[0] = __builtin_amdgcn_exp2f(s[kb][t * 8 + e]);
            pv[1] = __builtin_amdgcn_exp2f(s[kb][t * 8 + e + 1]);
            const half2v ph = __builtin_convertvector(pv, half2v);
            ls += pv[0] + pv[1];
            pf[kb][t][e] = ph[0];
            pf[kb][t][e + 1] = ph[1];
          }
    }
    l_i += ls;

    // ---- O^T += V^T P^T
    const char* vs = sb + KS_BYTES;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const char* vrow = vs + (db * 32 + ql) * 128;      // (row >> 1) & 7 == kswz for row = db*32 + ql
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const half8 vf = *(const half8*)(vrow + ((((kb * 2 + t) * 2 + hh) ^ kswz) << 4));
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kb][t], o[db], 0, 0, 0);
        }
    }
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of tile kt+1 landed; the barrier covers the others'
    __syncthreads();
    st_cur ^= 1;
  }

  const float l_tot = l_i + __shfl_xor(l_i, 32, 64);
  const int qr = q0 + ql;
  // ---- normalise and store O[q][head*64 + d]
  const float inv = 1.0f / l_tot;
  if (qr < p.Nq) {
    half_t* op = (half_t*)p.O + ((size_t)img * p.q_rows_per_img + qr) * p.ldo + head * 64 + 4 * hh;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        half4 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = (half_t)(o[db][g * 4 + e] * inv);
        *(half4*)(op + db * 32 + g * 8) = h;
      }
  }
}

// ======================================================================================================================================
// The software-pipelined kernel (pre-scaled Q).  Per wave and iteration t, ONE scheduling region holds
//     8 MFMAs  S(t+1) = K(t+1) Q^T - m       (K fragments from the stage of tile t+1)
//     8 MFMAs  O^T   += V^T(t) P(t)^T        (V^T fragments from the stage of tile t)
//     VALU     P(t) = exp2(S(t)) -> fp16 pairs (32 v_exp_f32, 16 v_cvt_pk_f16_f32), the row sum, and the row maximum of S(t+1)
// i.e. the scores are produced one iteration before they are consumed, so every MFMA has independent VALU work of the same wave to
// run beside (the guide's att[2] double pipeline, cdna_hip_programming.md T15), instead of the one-tile form's chain
// Q K^T -> max -> exp -> pack -> P V that leaves the overlap to whatever the other waves of the SIMD happen to be doing.
// VALU work per tile is also cut (the kernel is VALU-issue bound: 4 cycles per plain wave64 VALU instruction, 8 per v_exp_f32, against
// 32 per MFMA):
//   OPT bit 0  the running offset m enters as the C operand of the first Q K^T MFMA from a persistent 16-register splat that only the
//              rare rescale path touches (the one-tile form rebuilt it with 16 v_mov per tile);
//   OPT bit 1  row sums on the matrix pipe: v_mfma_f32_4x4x4_16b_f16 with A = ones gives every lane the sum of ITS OWN four B halves
//              (16 blocks of 4 lanes; D[b][i][j] = sum_k A[b][i][k] B[b][k][j]) -- 8 two-pass MFMAs per tile instead of 32 v_add_f32,
//              and the sum is over the fp16-ROUNDED probabilities, i.e. exactly what P V multiplies;
//   the partner-lane exchange of the row maximum is a v_permlane32_swap (VALU) instead of ds_bpermute + lgkmcnt(0), which also waited
//   for every fragment read in flight.
// Stage s of the LDS ring holds {K(t), V^T(t)} for t & 1 == s.  Iteration t reads K(t+1) and V^T(t); at its top it issues the DMA of
// K(t+2) into stage t & 1 (K(t) was last read in iteration t-1) and of V^T(t+1) into stage (t+1) & 1 (V^T(t-1) likewise); one
// vmcnt(0) + barrier per iteration, as before.
template <int NW, int OPT>
__global__ __launch_bounds__(NW * 64, UD_ATTN_PIPE_MINW) void attention_pipe_kernel(const UdAttention p, const float defer_thr) {
  constexpr bool NEGMC = (OPT & 1) != 0, MSUM = (OPT & 2) != 0;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5;
  const int ql = lane & 31;
  const int qt = (p.Nq + NW * 32 - 1) / (NW * 32);
  const int pairs = p.B * p.H;
  const int nt = (p.Nk + KT - 1) / KT;
  // PERSISTENT workgroups over work items (one item = one q-tile of one (image, head) pair).  XCD-aware: workgroup b runs on XCD b % 8 and walks
  // the items of the pairs pr % 8 == b % 8 with stride S = gridDim.x / 8, so all q-tiles of a pair stay on one XCD's L2.  With gridDim.x = the
  // number of items every workgroup has exactly one item (the round-4 launch shape); with 2 workgroups per CU (512) each walks ~2.75 items at the
  // encoder shape and the per-item fixed cost -- kernel-argument and Q loads, the first K / V^T DMAs, the output stores, the workgroup hand-over:
  // ~20 k of a wave's 48 k cycles, profiles/r05_attn_pmc.txt -- is hidden: the NEXT item's Q fragments and first tiles are fetched under the
  // last tile of the current one.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int n_items = ((pairs - xcd + 7) >> 3) * qt;       // items on this XCD
  int item = slot;
  if (item >= n_items) return;
  int head, img, kimg, q0;
  auto decode = [&](int it, int& hd, int& im, int& ki, int& qq) {
    const int pr = xcd + 8 * (it / qt);
    hd = pr % p.H;
    im = pr / p.H;
    ki = p.kv_broadcast ? (p.kv_group > 0 ? im / p.kv_group : 0) : im;
    qq = (it % qt) * (NW * 32) + wv * 32;
  };
  decode(item, head, img, kimg, q0);

  const half_t* Q = (const half_t*)p.Q;
  const half_t* K = (const half_t*)p.K;
  auto q_ptr = [&](int im, int hd, int qq) {
    int qr = qq + ql;
    qr = qr < p.Nq ? qr : p.Nq - 1;
    return Q + ((size_t)im * p.q_rows_per_img + qr) * p.ldq + hd * 64 + hh * 8;
  };
  auto make_rk = [&](int ki, int hd) { return ud_make_rsrc(K + (size_t)ki * p.k_rows_per_img * p.ldk + hd * 64, (unsigned)((p.Nk - 1) * p.ldk + 64) * 2u); };
  auto make_rv = [&](int ki, int hd) { return ud_make_rsrc((const half_t*)p.Vt + ((size_t)ki * p.H + hd) * 64 * (size_t)p.kv_ld, 64u * (unsigned)p.kv_ld * 2u); };

  half8 qf[4], qn[4];                                      // Q fragments of the current item / prefetched for the next one
  {
    const half_t* qp = q_ptr(img, head, q0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = *(const half8*)(qp + ks * 16);
      qn[ks] = qf[ks];
    }
  }

  const int lrow = lane >> 3, lch = lane & 7;
  ud_rsrc_t rK = make_rk(kimg, head), rV = make_rv(kimg, head);
  ud_rsrc_t rKn = rK, rVn = rV;                            // descriptors of the next item's pair
  constexpr int PW = 8 / NW;
  unsigned koff[PW], voff[PW];
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int row = (wv * PW + i) * 8 + lrow;
    const int ch = lch ^ ((row >> 1) & 7);
    koff[i] = (unsigned)(row * p.ldk + ch * 8) * 2u;
    voff[i] = (unsigned)(row * p.kv_ld + ch * 8) * 2u;
  }
  const unsigned kstep = (unsigned)(KT * p.ldk) * 2u;
  // a K tile beyond the sequence reads as zeros (descriptor bound) and a V^T tile beyond it is never consumed: no branch around the issue
  auto issue_k_from = [&](const ud_rsrc_t r, int kt, int stage) {
#pragma unroll
    for (int i = 0; i < PW; ++i) ud_bufl16(r, koff[i] + (unsigned)kt * kstep, 0, smem + stage * STAGE + (wv * PW + i) * 1024);
  };
  auto issue_v_from = [&](const ud_rsrc_t r, int kt, int stage) {
#pragma unroll
    for (int i = 0; i < PW; ++i) ud_bufl16(r, voff[i], kt * (KT * 2), smem + stage * STAGE + KS_BYTES + (wv * PW + i) * 1024);
  };
  auto issue_k = [&](int kt, int stage) { issue_k_from(rK, kt, stage); };
  auto issue_v = [&](int kt, int stage) { issue_v_from(rV, kt, stage); };

  f32x16 o[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.0f;
  float m_i = 0.0f, l_i = 0.0f;
  f32x4 la[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};     // MSUM: every component = the lane's own running row sum (low / high half4 of each fragment)
  f32x16 negm;                                                     // NEGMC: -m in every register
#pragma unroll
  for (int r = 0; r < 16; ++r) negm[r] = 0.0f;
  if constexpr (NEGMC) asm volatile("" : "+v"(negm));              // opaque: the optimiser must not rebuild the splat per tile

  const int kswz = (ql >> 1) & 7;
  // lane-constant LDS byte offsets of the four fragment chunks (the XOR swizzle is not additive, so one address per k-step)
  unsigned fo[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) fo[ks] = (unsigned)(ql * 128 + (((ks * 2 + hh) ^ kswz) << 4));
  const bool tail = (p.Nk & (KT - 1)) != 0;

  // Fragment i of an iteration, i = 0..15: 0..7 = K(t+1) fragments in Q K^T order (i = ks * 2 + kb), 8..15 = V^T(t) fragments in P V order
  // (i - 8 = (kb * 2 + g) * 2 + db).  They pass through a ring of FD registers sets, read FD - 1 fragments ahead of the MFMA that uses them.
  constexpr int FD = UD_ATTN_PIPE_FD;
  half8 fr[FD];
  f32x16 sc[2], sn[2];                                     // scores of tile t (being exponentiated) and of tile t + 1 (being produced)
  unsigned pw[16];                                         // P(t) as packed fp16 pairs: pw[a] = (P[2a], P[2a+1]) of the lane's 32 scores
  float mt, ls = 0.0f;

  auto frag_ptr = [&](const int i, const int stg_k, const int stg_v) -> const char* {
    return i < 8 ? smem + stg_k * STAGE + (i & 1) * 4096 + fo[i >> 1]
                 : smem + stg_v * STAGE + KS_BYTES + ((i - 8) & 1) * 4096 + fo[(i - 8) >> 1];
  };
  auto mask_tail = [&](f32x16 (&s)[2]) {                   // the last tile's keys beyond Nk
    const int kbase = (nt - 1) * KT + 4 * hh;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kbase + kb * 32 + (r & 3) + 8 * (r >> 2);
        if (key >= p.Nk) s[kb][r] = -1.0e30f;
      }
  };

  auto rowmax = [&](const f32x16 (&s)[2]) {
    float mx = s[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, s[kb][r]), s[kb][r + 1]);
    // partner lane ^ 32 without LDS: swap the upper half of one copy with the lower half of another
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
    return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
  };
  // Q K^T MFMA i (0..7) from ring slot i % FD into s
  auto mm_qk = [&](auto I_, f32x16 (&s)[2]) {
    constexpr int i = decltype(I_)::value, ks = i >> 1, kb = i & 1;
    if constexpr (ks == 0) {
      if constexpr (NEGMC) {
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[i % FD], qf[0], negm, 0, 0, 0);
      } else {
        f32x16 c0;
#pragma unroll
        for (int r = 0; r < 16; ++r) c0[r] = -m_i;
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[i % FD], qf[0], c0, 0, 0, 0);
      }
    } else {
      s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[i % FD], qf[ks], s[kb], 0, 0, 0);
    }
  };
  // exp2 + pack of score pair a (0..15) of sc
  auto ex = [&](auto A_) {
    constexpr int a = decltype(A_)::value;
    f32x2 pv;
    pv[0] = __builtin_amdgcn_exp2f(sc[a >> 3][(a & 7) * 2]);
    pv[1] = __builtin_amdgcn_exp2f(sc[a >> 3][(a & 7) * 2 + 1]);
    if constexpr (!MSUM) ls += pv[0] + pv[1];
    const half2v ph = __builtin_convertvector(pv, half2v);          // v_cvt_pk_f16_f32 (round to nearest even)
    pw[a] = __builtin_bit_cast(unsigned, ph);
  };
  // row-sum MFMA i (0..7) over the packed pairs 2i, 2i+1
  auto sm = [&](auto I_) {
    constexpr int i = decltype(I_)::value;
    if constexpr (MSUM) {
      typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
      const u32x2 w = {pw[2 * i], pw[2 * i + 1]};
      const half4 ones4 = {(half_t)1.0f, (half_t)1.0f, (half_t)1.0f, (half_t)1.0f};
      la[i & 1] = __builtin_amdgcn_mfma_f32_4x4x4f16(ones4, __builtin_bit_cast(half4, w), la[i & 1], 0, 0, 0);
    }
  };
  // P V MFMA j (0..7): fragment j + 8 from the ring, B operand = packed pairs 4 * (j >> 1) .. + 3
  auto mm_pv = [&](auto J_) {
    constexpr int j = decltype(J_)::value, db = j & 1, kg = j >> 1;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const u32x4 w = {pw[4 * kg], pw[4 * kg + 1], pw[4 * kg + 2], pw[4 * kg + 3]};
    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[(j + 8) % FD], __builtin_bit_cast(half8, w), o[db], 0, 0, 0);
  };

  // ---- the first item's K(0), V^T(0) -> stage 0, K(1) -> stage 1 (later items: prefetched under the previous item's last tile)
  issue_k(0, 0);
  issue_v(0, 0);
  issue_k(1, 1);
  bool has_next = false, v0_pending = false;
  int head_n = 0, img_n = 0, kimg_n = 0, q0_n = 0;
  for (;;) {                                               // ======================= one work item per trip =======================
  has_next = item + nslot < n_items;
  if (has_next) {
    decode(item + nslot, head_n, img_n, kimg_n, q0_n);
    rKn = make_rk(kimg_n, head_n);
    rVn = make_rv(kimg_n, head_n);
  }
  if (v0_pending) {                                        // odd tile count: the last tile read stage 0's V^T part, so V^T(0) could not be prefetched
    __builtin_amdgcn_s_barrier();                          // every wave is done with that tile
    issue_v(0, 0);
    v0_pending = false;
  }
  // ---- scores and row maximum of tile 0
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  {
    using namespace std;
#pragma unroll
    for (int i = 0; i < FD - 1; ++i) fr[i] = *(const half8*)frag_ptr(i, 0, 0);
    auto step = [&](auto I_) {
      constexpr int i = decltype(I_)::value;
      if constexpr (i + FD - 1 < 8) fr[(i + FD - 1) % FD] = *(const half8*)frag_ptr(i + FD - 1, 0, 0);
      mm_qk(I_, sc);
    };
    step(integral_constant<int, 0>{}); step(integral_constant<int, 1>{}); step(integral_constant<int, 2>{}); step(integral_constant<int, 3>{});
    step(integral_constant<int, 4>{}); step(integral_constant<int, 5>{}); step(integral_constant<int, 6>{}); step(integral_constant<int, 7>{});
  }
  if (nt == 1 && tail) mask_tail(sc);
  mt = rowmax(sc);
  // every wave has read its K(0) fragments before iteration 0 lets the DMA of K(2) into that stage (the fragment reads above were consumed by MFMAs,
  // i.e. they have returned, when a wave arrives here)
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();

  // KIND 0: steady tile; 1: tile nt-2 (the scores it produces are the last tile's: key-tail mask); 2: tile nt-1 (no next tile)
  auto body = [&](const int t, auto STG_, auto KIND_) {
    constexpr int STG = decltype(STG_)::value;
    constexpr int KIND = decltype(KIND_)::value;
    constexpr bool LAST = KIND == 2;
    // (1) rare: the row maximum of tile t exceeds the running offset by more than the threshold (always on the first tile)
    if (t == 0 || __any(mt > defer_thr)) {
      const float d = t == 0 ? mt : fmaxf(mt, 0.0f);
      m_i += d;
      if (t != 0) {                                        // first tile: O and l are still zero (and exp2(-d) may overflow)
        const float alpha = __builtin_amdgcn_exp2f(-d);
        l_i *= alpha;
        if constexpr (MSUM) {
          la[0] *= alpha;
          la[1] *= alpha;
        }
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[kb][r] -= d;
      if constexpr (NEGMC) {
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[r] -= d;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // (2) the first fragments (their LDS latency runs under the exponentials below), then the DMA of the tiles the NEXT iteration reads
    constexpr int F0 = LAST ? 8 : 0;                       // the last tile has no Q K^T half
#pragma unroll
    for (int i = F0; i < F0 + FD - 1; ++i) fr[i % FD] = *(const half8*)frag_ptr(i, STG ^ 1, STG);
    if constexpr (!LAST) {
      issue_k(t + 2, STG);
      issue_v(t + 1, STG ^ 1);
    } else if (has_next) {
      // the next item's first tiles and Q fragments, under this item's last tile (which reads only stage STG's V^T part)
      issue_k_from(rKn, 0, 0);
      issue_k_from(rKn, 1, 1);
      if constexpr (STG == 1) issue_v_from(rVn, 0, 0);
      else v0_pending = true;
      const half_t* qp = q_ptr(img_n, head_n, q0_n);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) qn[ks] = *(const half8*)(qp + ks * 16);
    }
    ls = 0.0f;
    // (3) sixteen slots, each {fragment read FD-1 ahead, one MFMA, a share of the VALU work}, pinned in this order by sched_barrier:
    //     slots 0..7  S(t+1) += K frag x Q frag        slots 8..15  O^T += V^T frag x P(t) quarter
    //     beside them, one score PAIR per slot 0..13 (2 v_exp_f32 + 1 v_cvt_pk_f16_f32; pairs 0 and 1 run ahead of slot 0, under the latency of the
    //     first fragment reads), a row-sum MFMA every other slot, and from slot 8 on the row maximum of S(t+1), four scores per slot:
    //     ~32 cycles of VALU issue beside every 32-cycle MFMA
    using namespace std;
    float mx = 0.0f;
    ex(integral_constant<int, 0>{});
    ex(integral_constant<int, 1>{});
    __builtin_amdgcn_sched_barrier(0);
    auto slot = [&](auto I_) {
      constexpr int i = decltype(I_)::value;
      if constexpr (i + FD - 1 < 16 && i + FD - 1 >= F0 + FD - 1) fr[(i + FD - 1) % FD] = *(const half8*)frag_ptr(i + FD - 1, STG ^ 1, STG);
      if constexpr (i < 8) {
        if constexpr (!LAST) mm_qk(I_, sn);
      } else {
        mm_pv(integral_constant<int, i - 8>{});
      }
      // row-sum MFMA k over pairs 2k, 2k+1 (packed in slots 2k-2, 2k-1): issued in slot 2k+1, well behind the v_cvt that wrote its operand
      if constexpr (i & 1) sm(integral_constant<int, (i / 2)>{});
      if constexpr (i < 14) ex(integral_constant<int, i + 2>{});
      // row maximum of S(t+1): complete after slot 7; block kb = 0 was finished by slot 6, so its half starts in slot 8
      if constexpr (!LAST && KIND == 0 && i >= 8) {
        constexpr int kb = (i - 8) >> 2, r0 = ((i - 8) & 3) * 4;
        if constexpr (i == 8) mx = sn[0][0];
        mx = fmaxf(fmaxf(mx, sn[kb][r0]), sn[kb][r0 + 1]);
        mx = fmaxf(fmaxf(mx, sn[kb][r0 + 2]), sn[kb][r0 + 3]);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    slot(integral_constant<int, 0>{}); slot(integral_constant<int, 1>{}); slot(integral_constant<int, 2>{}); slot(integral_constant<int, 3>{});
    slot(integral_constant<int, 4>{}); slot(integral_constant<int, 5>{}); slot(integral_constant<int, 6>{}); slot(integral_constant<int, 7>{});
    slot(integral_constant<int, 8>{}); slot(integral_constant<int, 9>{}); slot(integral_constant<int, 10>{}); slot(integral_constant<int, 11>{});
    slot(integral_constant<int, 12>{}); slot(integral_constant<int, 13>{}); slot(integral_constant<int, 14>{}); slot(integral_constant<int, 15>{});
    if constexpr (!MSUM) l_i += ls;
    if constexpr (!LAST) {
      if constexpr (KIND == 1) {
        if (tail) mask_tail(sn);
        mt = rowmax(sn);
      } else {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mt = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      }
      // (4) tiles of the next iteration landed (this wave's pieces; the barrier covers the others')
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) sc[kb] = sn[kb];
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  int t = 0;
  for (; t + 4 <= nt; t += 2) {                            // steady tiles (t + 1 <= nt - 3), two per trip: every LDS address is base + immediate
    body(t, I0{}, I0{});
    body(t + 1, I1{}, I0{});
  }
  const int rem = nt - t;                                  // 1 .. 3 tiles left; t is even
  if (rem == 3) {
    body(t, I0{}, I0{});
    body(t + 1, I1{}, I1{});
    body(t + 2, I0{}, I2{});
  } else if (rem == 2) {
    body(t, I0{}, I1{});
    body(t + 1, I1{}, I2{});
  } else {
    body(t, I0{}, I2{});
  }

  // ---- normalise and store O[q][head*64 + d]
  if constexpr (MSUM) l_i += la[0][0] + la[1][0];
  const float l_tot = l_i + __shfl_xor(l_i, 32, 64);
  const int qr = q0 + ql;
  const float inv = 1.0f / l_tot;
  if (qr < p.Nq) {
    half_t* op = (half_t*)p.O + ((size_t)img * p.q_rows_per_img + qr) * p.ldo + head * 64 + 4 * hh;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        half4 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = (half_t)(o[db][g * 4 + e] * inv);
        *(half4*)(op + db * 32 + g * 8) = h;
      }
  }
  if (!has_next) break;
  // ---- roll to the next item: its Q fragments, descriptors and first tiles are already on their way
  item += nslot;
  head = head_n; img = img_n; kimg = kimg_n; q0 = q0_n;
  rK = rKn; rV = rVn;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = qn[ks];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.0f;
  m_i = 0.0f; l_i = 0.0f;
  la[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; la[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 16; ++r) negm[r] = 0.0f;
  if constexpr (NEGMC) asm volatile("" : "+v"(negm));
  }                                                        // item loop
}

}  // namespace

extern "C" int ud_attention_f16(const UdAttention* desc, void* stream) {
  const UdAttention& d = *desc;
  if (!d.Q || !d.K || !d.Vt || !d.O || d.B <= 0 || d.H <= 0 || d.Nq <= 0 || d.Nk <= 0 || (d.ldq & 7) || (d.ldk & 7) ||
      (d.ldo & 3) || (d.kv_ld & 63) || d.kv_ld < ((d.Nk + 63) & ~63)) {
    ud_set_error("ud_attention_f16: bad argument (ldq/ldk % 8, kv_ld % 64, kv_ld >= roundup(Nk, 64))");
    return UD_ERR_BAD_ARG;
  }
  const int pairs = d.B * d.H;
  const float thr = (ud_debug_flags_host() & 1) ? -1.0f : 8.0f;
  if (UD_ATTN_PIPE != 0 && d.q_prescaled) {
    constexpr int NW = UD_ATTN_PIPE_NW;
    const int qt = (d.Nq + NW * 32 - 1) / (NW * 32);
    // persistent workgroups: UD_ATTN_PIPE_WGS_PER_XCD slots per XCD (64 = two 256-thread workgroups on each of an XCD's 32 CUs), every slot
    // walks the XCD's items with that stride; fewer items than slots: one workgroup per item
    const int items_xcd = ((pairs + 7) / 8) * qt;
    const int nslot = items_xcd < UD_ATTN_PIPE_WGS_PER_XCD ? items_xcd : UD_ATTN_PIPE_WGS_PER_XCD;
    hipLaunchKernelGGL((attention_pipe_kernel<NW, UD_ATTN_PIPE_OPT>), dim3(8 * nslot), dim3(NW * 64), 0, (hipStream_t)stream, d, thr);
  } else {
    const int qt = (d.Nq + 127) / 128;
    const dim3 grid(8 * ((pairs + 7) / 8) * qt);
    if (d.q_prescaled) hipLaunchKernelGGL((attention_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, d, thr);
    else hipLaunchKernelGGL((attention_kernel<0>), grid, dim3(256), 0, (hipStream_t)stream, d, thr);
  }
  UD_CHECK_LAUNCH("ud_attention_f16 launch");
  return UD_OK;
}
