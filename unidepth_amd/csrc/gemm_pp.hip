// Ping-pong form of the large-tile GEMM for the fp32 residual-accumulate class (round 6):  out(fp32)[M, N] (+)= A[M, K] W[N, K]^T + bias,
// optional fp16 copy, optional LayerNorm row statistics (UdGemm.row_stats_out / row_stats_final) -- the encoder's proj / fc2 launches
// (reference metadinov2/attention.py:60-62, mlp.py:38-40 with the residual add of block.py:85-89) and the decoder's 1x1 convolutions.
//
// Same tile, LDS image, MFMA orientation, K order and epilogue arithmetic as gemm256_kernel<3, UD_EPI_F32, dense, W3> (gemm.hip): every output
// element carries the SAME BITS (tests/test_kernels_gpu.py::test_gemm_ping_pong_is_bit_identical).  What differs is the schedule of a K-tile:
//   * 192 x 256 tile, 8 waves = 2 (m) x 4 (n); a K-tile is walked as TWO phases, one m-half (3 x 4 MFMA tiles x 2 k-steps = 24 MFMAs) of the
//     wave's 96 x 64 sub-tile each; a phase = { fragment reads + operand DMA issue ; s_barrier ; the MFMAs ; s_barrier }  (first half of round 6: four
//     phases of one quadrant; two measured -2 % on the proj / fc2 shapes, tools/ubench/gemm8p UD_PH2, profiles/r06_kloop_ablation.txt);
//   * the waves of m-row 1 run ONE barrier behind those of m-row 0, so on every SIMD one wave is inside its MFMA cluster while its partner
//     issues fragment reads / DMA (MI355X_MICROARCH "Two waves per SIMD"): the matrix pipe neither waits for an LDS round trip nor sees two
//     MFMA streams competing.  tools/ubench/gemm8p.hip, same process, interleaved (profiles/r06_gemm_loop_diagnostic.txt): +6.5 % on the proj
//     shape, +3 % on the fc2 shape, +6 % at 4096^3 against the product's one-barrier-per-K-tile loop; on multi-round launches (qkv, fc1) the
//     product's continuous K-tile stream across tiles wins, so this kernel takes ONE-ROUND tile lists only;
//   * two K-tile buffers (112 KB): the activation operand runs one K-tile ahead, the weight operand two -- W(kt + 2) re-fills W(kt)'s region
//     in phase 2, its four DMA instructions stay in flight across the K-tile boundary (counted vmcnt(4), never 0 in the loop).
//     DMA of K-tile kt (buffer b = kt & 1):  phase 1: A(kt+1) -> b^1     phase 2: W(kt+2) -> b ; vmcnt(4)
//     WAR: A(b^1) was last read in phase 2 of kt-1 and W(b) in phase 1 of kt; both sets of reads are retired (lgkmcnt(0)) BEFORE the barrier that
//     follows them, and the wave row that re-fills the region issues at least one barrier later.  RAW: a wave waits for its own DMA (vmcnt) before
//     a barrier every reader passes.
#include "ud_common.h"
#include <cstdlib>

namespace {

template <int I> struct IntTag { static constexpr int value = I; };
template <bool B> struct BoolTag { static constexpr bool value = B; };
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ unsigned pp_pack2(float x, float y) {
  f32x2 v; v[0] = x; v[1] = y;
  const half2v h = __builtin_convertvector(v, half2v);
  return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ void pp_pair16(unsigned& a, unsigned& b) {
  const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}
// the row-statistics arithmetic of gemm.hip (ud_row_stats_acc / ud_row_stats_store), restated operation for operation: same bits
__device__ __forceinline__ void pp_stats_acc(const f32x4 v, float& s1, float& s2) {
  s1 += (v[0] + v[1]) + (v[2] + v[3]);
  s2 = __builtin_fmaf(v[0], v[0], s2);
  s2 = __builtin_fmaf(v[1], v[1], s2);
  s2 = __builtin_fmaf(v[2], v[2], s2);
  s2 = __builtin_fmaf(v[3], v[3], s2);
}
__device__ __forceinline__ void pp_stats_store(const UdGemm& p, float s1, float s2, int m, int nbase, int lane, bool ok) {
  s1 += __shfl_xor(s1, 16, 64);
  s2 += __shfl_xor(s2, 16, 64);
  s1 += __shfl_xor(s1, 32, 64);
  s2 += __shfl_xor(s2, 32, 64);
  if (ok && (lane >> 4) == 0) {
    f32x2 o;
    o[0] = s1; o[1] = s2;
    float* dst = p.row_stats_out + ((size_t)m * (p.N >> 6) + (nbase >> 6)) * 2;
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_nop 2" ::"v"(dst), "v"(o) : "memory");
  }
}

#define PP_BAR()                             \
  do {                                       \
    __builtin_amdgcn_sched_barrier(0);       \
    __builtin_amdgcn_s_barrier();            \
    __builtin_amdgcn_sched_barrier(0);       \
  } while (0)

// NW = 8: one workgroup of 8 waves (2 x 4) per CU on a 64 MQ x 256 tile.
// NW = 4 ("duo"): 4 waves (2 x 2) on a 64 MQ x 128 tile, 80 KB of LDS and <= 256 registers per lane, so that TWO workgroups share a
// CU -- one wave of each on every SIMD (tools/ubench/placement.hip, profiles/r06_placement.txt: workgroups j and j + 32 of an XCD's dispatch order land on
// the same CU, both resident from the start; 81 920 B of LDS is exactly half a CU, one byte more and the second workgroup waits for the first).  The
// per-wave code (96 x 64 sub-tile, fragment addresses, K order, epilogue) is the 8-wave kernel's: same bits.  The two workgroups of a CU are independent
// instruction streams: the second-dispatched one runs at lower priority (prio_mode), so it falls behind the first and its K loop covers the first one's
// store burst -- the only form in which one tile's bursts overlap another tile's K loop when a launch is a single round of tiles.
template <int MQ, int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_pp_f32_kernel(const UdGemm p, const int prio_mode) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = 64 * MQ;
  constexpr int BN = NW * 32;                   // 4 (8 waves) or 2 (4 waves) wave columns of 64
  constexpr int NWC = NW / 2;
  constexpr int TMC = 2 * MQ;
  constexpr int A_BYTES = BM * 128;
  constexpr int BUFB = A_BYTES + BN * 128;
  constexpr int RPI = NW * 8;                   // tile rows one DMA instruction of the workgroup covers (16 bytes per lane, 8 lanes per row)
  constexpr int A_LD = BM / RPI;                // DMA instructions per thread per K-tile of A
  constexpr int W_LD = BN / RPI;                // ... of W (4 for both forms: the counted waits below say 4)
  static_assert(W_LD == 4, "the counted vmcnt waits assume four W instructions per K-tile");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wv / NWC, wc = wv % NWC;
  const int nk = p.K >> 6;                      // even, >= 4 (launch condition)
  const int tiles_n = p.N / BN, tiles_m = (p.M + BM - 1) / BM;
  const int nblk = tiles_m * tiles_n;
  if (NW == 4) {
    // static priority per workgroup (s_setprio is a scalar instruction: branch on a uniform value)
    const bool second = (int)(blockIdx.x >> 3) >= 32;
    if (prio_mode == 1) { if (second) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(2); }
    else if (prio_mode == 2) { if (second) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0); }
  }
  const ud_rsrc_t rA = ud_make_rsrc(p.A, 0x80000000u), rW = ud_make_rsrc(p.W, 0x80000000u);
  const int lrow = tid >> 3;
  const int csrc = (tid & 7) ^ ((lrow >> 1) & 7);
  const int fswz = (lane & 15) >> 1;
  const int c0 = ((lane >> 4) ^ fswz) << 4, c1 = ((4 + (lane >> 4)) ^ fswz) << 4;
  const int a_off = (wr * (BM / 2) + (lane & 15)) * 128;
  const int b_off = A_BYTES + (wc * 64 + (lane & 15)) * 128;

  for (int t = blockIdx.x; t < nblk; t += gridDim.x) {
    // tile list entry -> (m0, n0): the large-tile kernel's map (gemm.hip decode()): a contiguous range of the list per XCD, walked row-major
    // when the list fits one round, in groups of 8 row tiles otherwise
    int m0, n0;
    {
      const int q = nblk >> 3, r = nblk & 7;
      const int xcd = t & 7, idx = t >> 3;
      const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
      const int GM = nblk <= (NW == 8 ? 256 : 512) ? 1 : 8;
      const int gsz = GM * tiles_n;
      const int grp = bid / gsz;
      const int first_m = grp * GM;
      const int gm = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
      const int rem = bid - grp * gsz;
      m0 = (first_m + rem % gm) * BM;
      n0 = (rem / gm) * BN;
    }
    unsigned va[A_LD], vb[W_LD];                    // rows past M re-read the last row (their outputs are never stored)
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      int m = m0 + lrow + RPI * j;
      m = m < p.M ? m : p.M - 1;
      va[j] = ((unsigned)m * (unsigned)p.lda + csrc * 8) * 2u;
    }
#pragma unroll
    for (int j = 0; j < W_LD; ++j) vb[j] = ((unsigned)(n0 + lrow + RPI * j) * (unsigned)p.ldw + csrc * 8) * 2u;
    auto issueA = [&](int kt, int buf, auto J0, auto J1) {
      char* sb = smem + buf * BUFB + wv * 1024;
#pragma unroll
      for (int j = decltype(J0)::value; j < decltype(J1)::value; ++j) ud_bufl16(rA, va[j], kt * 128, sb + j * (RPI * 128));
    };
    auto issueB = [&](int kt, int buf, auto J0, auto J1) {
      char* sb = smem + buf * BUFB + A_BYTES + wv * 1024;
#pragma unroll
      for (int j = decltype(J0)::value; j < decltype(J1)::value; ++j) ud_bufl16(rW, vb[j], kt * 128, sb + j * (RPI * 128));
    };

    const int mbase = m0 + wr * (BM / 2), nbase = n0 + wc * 64;
    f32x4 acc[TMC][4];
#pragma unroll
    for (int i = 0; i < TMC; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    half8 a[MQ][2], b0[2][2], b1[2][2];

    // ---- prologue: A(0), W(0) -> buffer 0, W(1) -> buffer 1 (stays in flight), then the old fp32 values = the accumulators' initial state
    // (gemm.hip explains why: summation order independent of the row's position, batch-permutation equivariance stays bit-exact), issued in
    // the order the phases of K-tile 0 consume them: the compiler's counted waits before each quadrant's first MFMA leave the rest of the
    // burst in flight
    issueA(0, 0, IntTag<0>{}, IntTag<A_LD>{});
    issueB(0, 0, IntTag<0>{}, IntTag<4>{});
    issueB(1, 1, IntTag<0>{}, IntTag<4>{});
    asm volatile("" ::: "memory");               // the counted wait below needs this order: operand DMA first, the old values after it
    __builtin_amdgcn_sched_barrier(0);
    const bool full = m0 + BM <= p.M;
    bool counted = false;
    if (p.accumulate) {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const float* ob = (const float*)p.out + (size_t)(mbase + (ln & 15)) * p.ldc + nbase + 4 * (ln >> 4);
      if (full) {
#define PP_PRELOAD(MQI, NQI)                                                                    \
  _Pragma("unroll") for (int i = 0; i < MQ; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)  \
      acc[(MQI) * MQ + i][(NQI) * 2 + j] = *(const f32x4*)(ob + (size_t)(((MQI) * MQ + i) * 16) * p.ldc + ((NQI) * 2 + j) * 16);
        PP_PRELOAD(0, 0) PP_PRELOAD(0, 1) PP_PRELOAD(1, 1) PP_PRELOAD(1, 0)
#undef PP_PRELOAD
        counted = true;
      } else {
#pragma unroll
        for (int i = 0; i < TMC; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (mbase + i * 16 + (ln & 15) < p.M) acc[i][j] = *(const f32x4*)(ob + (size_t)(i * 16) * p.ldc + j * 16);
      }
    }
    if (counted) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(TMC * 4 + 4) : "memory");      // A(0), W(0) landed; W(1) and the old values in flight
    else if (p.accumulate) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    PP_BAR();
    if (wr == 1) PP_BAR();                       // m-row 1 runs one barrier behind m-row 0 from here on

#define PP_MFMA_QUAD(MQI, NQI, BF)                                                                                       \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int i = 0; i < MQ; ++i)                        \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[(MQI) * MQ + i][(NQI) * 2 + j] =                                 \
          __builtin_amdgcn_mfma_f32_16x16x32_f16(BF[j][ks], a[i][ks], acc[(MQI) * MQ + i][(NQI) * 2 + j], 0, 0, 0);

    // one K-tile; HEAD: K-tile 0 (both look-aheads exist: nk >= 4; no run-time conditions, so the compiler's counted waits for the
    // preloaded accumulators stay exact)
    auto ktile = [&](auto BUFT, auto HEADT, int kt) {
      constexpr int buf = decltype(BUFT)::value;
      constexpr bool HEAD = decltype(HEADT)::value;
      const char* sb = smem + buf * BUFB;
      const bool n1 = HEAD || kt + 1 < nk, n2 = HEAD || kt + 2 < nk;
      // ---- phase 1: quadrants (0, 0) and (0, 1) -- the wave's first m-half against all four column tiles.  (Round 6, second half: a K-tile was FOUR
      // phases of one quadrant each; two phases of two quadrants -- 4 barriers per K-tile instead of 8, 24 MFMAs between a pair -- measured
      // -2 % on the proj / fc2 shapes in tools/ubench/gemm8p, profiles/r06_kloop_ablation.txt.  Same MFMA order per accumulator: same bits.)
      // The W reads of this buffer end here: retired (lgkmcnt(0)) before the barrier, phase 2 re-fills the W region.
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        b0[j][0] = *(const half8*)(sb + b_off + j * 2048 + c0);
        b0[j][1] = *(const half8*)(sb + b_off + j * 2048 + c1);
      }
#pragma unroll
      for (int i = 0; i < MQ; ++i) {
        a[i][0] = *(const half8*)(sb + a_off + i * 2048 + c0);
        a[i][1] = *(const half8*)(sb + a_off + i * 2048 + c1);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        b1[j][0] = *(const half8*)(sb + b_off + (2 + j) * 2048 + c0);
        b1[j][1] = *(const half8*)(sb + b_off + (2 + j) * 2048 + c1);
      }
      if (n1) issueA(kt + 1, buf ^ 1, IntTag<0>{}, IntTag<A_LD>{});
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b1[0][0]), "+v"(b1[0][1]), "+v"(b1[1][0]), "+v"(b1[1][1]), "+v"(b0[0][0]), "+v"(b0[0][1]), "+v"(b0[1][0]), "+v"(b0[1][1])::"memory");
      PP_BAR();
      PP_MFMA_QUAD(0, 0, b0)
      PP_MFMA_QUAD(0, 1, b1)
      PP_BAR();
      // ---- phase 2: quadrants (1, 1) and (1, 0); the next K-tile's operands have landed (own DMA) before the barrier every reader passes
#pragma unroll
      for (int i = 0; i < MQ; ++i) {
        a[i][0] = *(const half8*)(sb + a_off + (MQ + i) * 2048 + c0);
        a[i][1] = *(const half8*)(sb + a_off + (MQ + i) * 2048 + c1);
      }
      if (n2) {
        issueB(kt + 2, buf, IntTag<0>{}, IntTag<4>{});
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      // the A reads of this buffer end here: retired before the barrier (the other wave row issues A(kt + 2) into it one barrier later)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PP_BAR();
      PP_MFMA_QUAD(1, 1, b1)
      PP_MFMA_QUAD(1, 0, b0)
      PP_BAR();
    };
    ktile(IntTag<0>{}, BoolTag<true>{}, 0);
    ktile(IntTag<1>{}, BoolTag<false>{}, 1);
    for (int kt = 2; kt < nk; kt += 2) {
      ktile(IntTag<0>{}, BoolTag<false>{}, kt);
      ktile(IntTag<1>{}, BoolTag<false>{}, kt + 1);
    }
#undef PP_MFMA_QUAD
    if (wr == 0) PP_BAR();                       // back in step: the epilogue's barriers mean "every wave"

    // =================================== epilogue (gemm256_kernel's straight-line fp32 path, rows past M masked) ===================================
    int eln = lane;
    asm volatile("" : "+v"(eln));
    const int frow = eln & 15, fq = eln >> 4;
    f32x4 bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = *(const f32x4*)(p.bias + nbase + j * 16 + 4 * fq);
    float* o = (float*)p.out + (size_t)(mbase + frow) * p.ldc + nbase + 4 * fq;
    half_t* o2 = p.out2 ? (half_t*)p.out2 + (size_t)(mbase + frow) * p.ldc2 + nbase + 16 * (fq & 1) + 8 * (fq >> 1) : nullptr;
    const bool wr32 = p.accumulate != 2;
    const bool lre = p.act2 == UD_ACT_LRELU;
    bool ticket_taken = false;
    unsigned ticket_val = 0;
    if (p.row_stats_final) {
      // partial sums first, then the ticket, then the big stores (gemm.hip: the ticket's round trip hides under the stores)
#pragma unroll
      for (int i = 0; i < TMC; ++i) {
        float rs1 = 0.f, rs2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) pp_stats_acc(acc[i][j] + bv[j], rs1, rs2);
        pp_stats_store(p, rs1, rs2, mbase + i * 16 + frow, nbase, eln, mbase + i * 16 + frow < p.M);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (tid == 0) ticket_val = atomicInc(p.row_stats_ticket + m0 / BM, (unsigned)tiles_n - 1u);
      ticket_taken = true;
    }
#pragma unroll
    for (int i = 0; i < TMC; ++i) {
      const bool mok = mbase + i * 16 + frow < p.M;
      unsigned w[4][2];
      float rs1 = 0.f, rs2 = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 v = acc[i][j] + bv[j];
        if (wr32 && mok) *(f32x4*)(o + (size_t)(i * 16) * p.ldc + j * 16) = v;
        pp_stats_acc(v, rs1, rs2);
        f32x4 av = v;
        if (lre) {
#pragma unroll
          for (int r = 0; r < 4; ++r) av[r] = ud_lrelu(v[r]);
        }
        w[j][0] = pp_pack2(av[0], av[1]);
        w[j][1] = pp_pack2(av[2], av[3]);
      }
      if (o2) {
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          pp_pair16(w[2 * jp][0], w[2 * jp + 1][0]);
          pp_pair16(w[2 * jp][1], w[2 * jp + 1][1]);
          u32x4 s;
          s[0] = w[2 * jp][0]; s[1] = w[2 * jp][1]; s[2] = w[2 * jp + 1][0]; s[3] = w[2 * jp + 1][1];
          if (mok) *(u32x4*)(o2 + (size_t)(i * 16) * p.ldc2 + jp * 32) = s;
        }
      }
      if (p.row_stats_out && !ticket_taken) pp_stats_store(p, rs1, rs2, mbase + i * 16 + frow, nbase, eln, mok);
    }
    if (p.row_stats_final) {
      // the LAST of the tiles_n workgroups of this row tile reduces the partial sums of all column tiles (ascending slab order), gemm.hip
      unsigned* flag = (unsigned*)(smem + (NW == 8 ? 2 * BUFB : 0));      // duo: no byte beyond the two buffers (all LDS reads are retired here)
      if (tid == 0) *flag = ticket_val;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const unsigned arrived = *flag;
      if (arrived == (unsigned)tiles_n - 1u) {
        const int slabs = p.N >> 6;
        if (tid < BM && m0 + tid < p.M) {
          const float* src = p.row_stats_out + (size_t)(m0 + tid) * slabs * 2;
          f32x4 q[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            q[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (2 * k < slabs) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(q[k]) : "v"(src + 4 * k) : "memory");
          }
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7])::"memory");
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            s1 += q[k][0]; s2 += q[k][1];
            s1 += q[k][2]; s2 += q[k][3];
          }
          const float inv = 1.0f / (float)p.ln_D;
          const float mean = s1 * inv;
          const float var = fmaxf(__builtin_fmaf(-mean, mean, s2 * inv), 0.0f);
          f32x2 oo;
          oo[0] = rsqrtf(var + p.ln_eps);
          oo[1] = -mean * oo[0];
          *(f32x2*)(p.row_stats_final + 2 * (size_t)(m0 + tid)) = oo;
        }
      }
    }
    // the next tile's prologue overwrites both buffers and the flag word: every wave is past its last LDS read (un-stagger barrier above;
    // the flag is read before the barrier-free tail of the finalizer)
    if (t + (int)gridDim.x < nblk) PP_BAR();
  }
}

}  // namespace

// host side: eligibility and launch (called from ud_gemm_f16, gemm.hip)
bool ud_gemm_pp_ok(const UdGemm& d) {
  if (d.epi != UD_EPI_F32 || d.amode != UD_A_DENSE || d.groups > 1 || !d.bias || d.add || d.rows_in || d.max_out || d.act != UD_ACT_NONE) return false;
  if (d.act2 != UD_ACT_NONE && d.act2 != UD_ACT_LRELU) return false;
  if ((d.N & 255) || (d.K & 127) || d.K < 256 || d.M < 1024 || d.a_wrap || d.w_wrap || d.row_stats_in || d.up_src) return false;
  if ((d.ldc & 3) || (d.out2 && (d.ldc2 & 7))) return false;
  if (2.0 * d.M * d.lda >= 2147483648.0 || 2.0 * d.N * d.ldw >= 2147483648.0) return false;
  if (d.row_stats_final && (!d.row_stats_out || !d.row_stats_ticket || d.ln_D <= 0 || d.N > 1024 || (d.N & 127))) return false;
  return true;
}

int ud_gemm_pp_launch(const UdGemm& d, hipStream_t s) {
  constexpr int MQ = 3;
  constexpr int LDS = 2 * (64 * MQ * 128 + 32768) + 64;
  const int tiles = (d.N >> 8) * ((d.M + 64 * MQ - 1) / (64 * MQ));
  static bool attr_set[UD_MAX_DEVICES];
  if (!ud_attr_once(attr_set)) {
    if (hipFuncSetAttribute((const void*)gemm_pp_f32_kernel<MQ, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
      ud_set_error("ud_gemm_f16: cannot reserve the LDS buffers of the ping-pong large-tile kernel");
      return UD_ERR_LAUNCH;
    }
  }
  hipLaunchKernelGGL((gemm_pp_f32_kernel<MQ, 8>), dim3(tiles < 256 ? tiles : 256), dim3(512), LDS, s, d, 0);
  UD_CHECK_LAUNCH("ud_gemm_f16 (large tile, ping-pong) launch");
  return UD_OK;
}

// the two-workgroups-per-CU form: 192 x 128 tiles, one round of at most 512
bool ud_gemm_duo_ok(const UdGemm& d) {
  if (!ud_gemm_pp_ok(d) || (d.N & 127)) return false;
  const int tiles = (d.N >> 7) * ((d.M + 191) / 192);
  return tiles <= 512;
}

int ud_gemm_duo_launch(const UdGemm& d, hipStream_t s, int prio_mode) {
  constexpr int MQ = 3;
  constexpr int LDS = 2 * (64 * MQ * 128 + 16384);          // 81 920 B: exactly half of a CU's LDS
  const int tiles = (d.N >> 7) * ((d.M + 64 * MQ - 1) / (64 * MQ));
  static bool attr_set[UD_MAX_DEVICES];
  if (!ud_attr_once(attr_set)) {
    if (hipFuncSetAttribute((const void*)gemm_pp_f32_kernel<MQ, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
      ud_set_error("ud_gemm_f16: cannot reserve the LDS buffers of the two-workgroups-per-CU kernel");
      return UD_ERR_LAUNCH;
    }
  }
  hipLaunchKernelGGL((gemm_pp_f32_kernel<MQ, 4>), dim3(tiles), dim3(256), LDS, s, d, prio_mode);
  UD_CHECK_LAUNCH("ud_gemm_f16 (two workgroups per CU) launch");
  return UD_OK;
}
