// Fused multi-head attention forward for gfx950 (flash style, online softmax), head_dim = 64, fp16 operands,
// fp32 scores / statistics / accumulators.  One workgroup = 4 waves x 32 query rows; K and V^T tiles of 64 keys go
// HBM/L2 -> LDS by global_load_lds_dwordx4 (no VGPR staging, no ds_write pass), double buffered: the DMA of tile t+1 is
// issued before the MFMAs of tile t and awaited (vmcnt(0) + barrier) after them.
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
#include "ud_common.h"
#include <type_traits>
#ifndef UD_ATTN_SPLIT_SM
#define UD_ATTN_SPLIT_SM 0
#endif
// Round-4 compile-time variants (tools/r4_attn_variants.sh builds one library per combination; DESIGN 10.2 has the measurements):
//   UD_ATTN_NSTAGE  K / V^T ring depth (2 = one tile ahead, awaited with vmcnt(0) at the end of every tile; 3 / 4 = two / three tiles
//                   ahead, the end-of-tile wait is a COUNTED vmcnt that leaves the younger tiles in flight, raw s_barrier)
//   UD_ATTN_NOPRIO  no s_setprio(1) around the P V MFMAs
//   UD_ATTN_OAGPR   the O^T accumulators pinned to AGPRs (inline-asm MFMA, "+a"): the P V MFMAs then read / write C and D through the
//                   accumulator file instead of the VGPR ports the other waves' VALU work needs
//   UD_ATTN_NOMAX   MODE 1: no per-tile row maximum.  P = exp2(S - m) is formed optimistically against the running offset m and the tile is
//                   redone on the slow path (exact maximum, rescale) only when a lane's partial row sum exceeds 2^15 -- every P <= 2^15 is
//                   exactly representable in fp16 range and the statistics are fp32.  The first tile of a workgroup always takes the slow path.
#ifndef UD_ATTN_NSTAGE
#define UD_ATTN_NSTAGE 2
#endif
#ifndef UD_ATTN_NOPRIO
#define UD_ATTN_NOPRIO 0
#endif
#ifndef UD_ATTN_OAGPR
#define UD_ATTN_OAGPR 0
#endif
#ifndef UD_ATTN_NOMAX
#define UD_ATTN_NOMAX 0
#endif
#ifndef UD_ATTN_UNROLL2
#define UD_ATTN_UNROLL2 0
#endif
#ifndef UD_ATTN_MFMASUM
#define UD_ATTN_MFMASUM 0          // NOMAX only: row sums by v_mfma_f32_4x4x4_16b_f16 (A = ones) instead of 32 v_add_f32 per tile
#endif
#ifndef UD_ATTN_QB
#define UD_ATTN_QB 1            // 32-row query blocks per WAVE (product path, MODE 1): 2 = every K / V^T fragment read from LDS feeds two MFMAs, and a staged
#endif                          // tile serves 256 query rows -- half the ds_reads, DMA instructions and barriers per FLOP at ~2x the registers (2 waves per SIMD)
#ifndef UD_ATTN_MINW
#define UD_ATTN_MINW 1          // minimum waves per SIMD promised to the register allocator (4 = cap the kernel at 128 VGPRs)
#endif

#ifndef UD_ATTN_TRACE
#define UD_ATTN_TRACE 0
#endif
#if UD_ATTN_TRACE
// tools build only (tools/r4_attn_trace.py): per-wave sums of the shader-clock time between seven points of the tile loop, added to
// ud_attn_trace_ptr[0..5] (+ [6] = wave-tiles counted) at the end of every wave.  s_memtime is an SMEM read, so every stamp also waits for
// the wave's outstanding LDS reads: the stamped build runs a few percent slower and shows WHERE a tile's time goes, not how long it takes.
__device__ unsigned long long* ud_attn_trace_ptr = nullptr;
extern "C" int ud_attn_trace_set(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(ud_attn_trace_ptr), &buf, sizeof(buf)) == hipSuccess ? UD_OK : UD_ERR_LAUNCH;
}
#define UD_ATT_STAMP(i)                                        \
  do {                                                         \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    if (i > 0) tsum[i - 1] += now_ - tprev;                    \
    tprev = now_;                                              \
  } while (0)
#else
#define UD_ATT_STAMP(i)
#endif

namespace {

constexpr int KT = 64;             // keys per tile
constexpr int KS_BYTES = 64 * 128;
constexpr int VS_BYTES = 64 * 128;
constexpr int STAGE = KS_BYTES + VS_BYTES;

// ABL: ablation mask for tools/ablate_attn.py (instrumented builds only; the product instantiates ABL = 0):
//   1 no exp/softmax VALU, 2 no P V MFMAs, 4 no Q K^T MFMAs, 8 no K/V tile traffic (global loads + LDS stores), 16 no barrier
// MODE 0: Q as given, P = exp2(S c - m c) with c = scale log2(e) (one fma per score).
// MODE 1 / 2: Q arrives PRE-SCALED by c (UdAttention.q_prescaled: the engine folds c into the q projection's weights at load time), and
//   the running maximum is subtracted by the MFMA itself -- the score accumulators start at -m instead of 0 -- so P = exp2(S'') with
//   no VALU in between: 31 fewer VALU instructions per 64-key tile and lane (the kernel is VALU-issue bound: PMC VALU active 53 %, MFMA
//   busy 34 %).  A growing maximum (deferred, threshold 2^8 as before) is handled on the rare path by shifting S''.
//   MODE 2 additionally forms the row sums with v_dot2_f32_f16 on the packed P pairs (16 instead of 32 adds; the sum is then over the
//   ROUNDED probabilities, i.e. exactly what P V multiplies).
// NW = waves per workgroup (4 or 8), 32 query rows each: the NW * 32 rows of a workgroup share every K / V^T tile it stages, so 8 waves
// halve the L2 -> LDS traffic and the DMA issue + barrier work per query row (the ablation of the 4-wave kernel put the K / V traffic at
// ~20 % of its time); the register budget (<= 128 VGPRs at 512 threads) is the 4-wave kernel's own 124.
// SPLIT: split-key mode (UdAttention.k_chunk / part): the workgroup covers ONE chunk of the keys and leaves its un-normalised accumulators,
// running maximum and row sum in `part`; attention_merge_kernel combines the chunks.  For few queries against many keys (the Nystrom
// kernel_3 product: 128 landmark queries x up to 19200 keys per (image, head) -- one workgroup per pair would walk 300 key tiles alone).
template <int ABL, int MODE, int NW = 4, bool SPLIT = false, int QB = 1>
__global__ __launch_bounds__(NW * 64, UD_ATTN_MINW) void attention_kernel(const UdAttention p, const float defer_thr) {
  static_assert(QB == 1 || (MODE == 1 && !SPLIT && !UD_ATTN_NOMAX && !UD_ATTN_SPLIT_SM && !UD_ATTN_OAGPR && ABL == 0), "QB > 1: the product MODE 1 path only");
  constexpr int NST = UD_ATTN_NSTAGE;
  __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5;
  const int ql = lane & 31;
  // XCD-aware work map (1-D grid; workgroup b runs on XCD b % 8): the q-tiles of one (image, head) pair share its K / V^T
  // (2 x 175 KB at N = 1370), so all of them go to ONE XCD, in consecutive dispatch slots -- with the natural 3-D grid they
  // were spread over all 8 private L2s and every XCD pulled nearly every K/V through the fabric (rocprofv3 FETCH_SIZE
  // 403 MB per launch against 67 MB of unique Q/K/V: the kernel ran at the fabric read rate, 4.1 TB/s, not at MFMA rate).
  const int qt = (p.Nq + NW * 32 * QB - 1) / (NW * 32 * QB);
  const int pairs = p.B * p.H;
  const int nt = (p.Nk + KT - 1) / KT;
  const int tpc = SPLIT ? p.k_chunk / KT : nt;            // key tiles per chunk
  const int nc = SPLIT ? (nt + tpc - 1) / tpc : 1;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int pr = xcd + 8 * (slot / (qt * nc));            // all q-tiles and chunks of a pair on one XCD, consecutive slots
  if (pr >= pairs) return;
  const int head = pr % p.H;
  const int img = pr / p.H;
  const int kimg = p.kv_broadcast ? (p.kv_group > 0 ? img / p.kv_group : 0) : img;
  const int rem = slot % (qt * nc);
  const int chunk = rem / qt;
  const int q0 = (rem % qt) * (NW * 32 * QB) + wv * (32 * QB);
  const int kt0 = chunk * tpc;
  const int kt1 = SPLIT ? (kt0 + tpc < nt ? kt0 + tpc : nt) : nt;

  const half_t* Q = (const half_t*)p.Q;
  const half_t* K = (const half_t*)p.K;
  const half_t* Vt = (const half_t*)p.Vt + ((size_t)kimg * p.H + head) * 64 * (size_t)p.kv_ld;

  // ---- Q fragments (B operand): q = ql, d = ks*16 + hh*8 .. +8
  half8 qf[QB][4];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    int qr = q0 + qb * 32 + ql;
    qr = qr < p.Nq ? qr : p.Nq - 1;
    const half_t* qp = Q + ((size_t)img * p.q_rows_per_img + qr) * p.ldq + head * 64 + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[qb][ks] = *(const half8*)(qp + ks * 16);
  }

  // ---- tile loader: 512 16-byte chunks per operand tile = 8 wave-instructions of 8 rows x 128 B; wave w issues pieces 2w, 2w+1.
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

  f32x16 oq[QB][2];
  float m_q[QB], l_q[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) oq[qb][db][r] = 0.0f;
    m_q[qb] = MODE ? 0.0f : -1.0e30f;
    l_q[qb] = 0.0f;
  }
  // the one-block forms below (variants, ablations, split-key mode) keep their names: block 0
  f32x16 (&o)[2] = oq[0];
  float& m_i = m_q[0];
  float& l_i = l_q[0];
  const float c = p.scale * 1.4426950408889634f;

  // ring prologue: tiles kt0 .. kt0 + NST - 2 go out, all of them awaited once (a few hundred ns per workgroup, off the per-tile path)
#pragma unroll
  for (int i = 0; i < NST - 1; ++i)
    if (kt0 + i < kt1) issue(kt0 + i, i);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int kswz = (ql >> 1) & 7;   // (key >> 1) & 7 for key = kb*32 + ql
  int st_cur = 0, st_new = NST - 1;   // ring slots of tile kt and of the tile issued during it (kt + NST - 1)
#if UD_ATTN_TRACE
  unsigned long long tsum[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
#endif
  // UD_ATTN_UNROLL2 (2-deep ring): the tile body is instantiated per ring slot and the loop walks two tiles per iteration, so every LDS
  // address of a tile is the lane's base + an immediate offset (no per-tile v_add of the slot offset: ~8 VALU instructions per tile)
  auto tile = [&](const int kt, auto SLOT) {
    constexpr int CS = decltype(SLOT)::value;               // >= 0: compile-time ring slot of this tile (NST == 2), -1: st_cur / st_new
    UD_ATT_STAMP(0);
    const bool ahead = kt + NST - 1 < kt1;
    if constexpr (!(ABL & 8)) {
      if (ahead) issue(kt + NST - 1, CS >= 0 ? (CS ^ 1) : st_new);   // every wave passed the barrier that ended tile kt-1, the last reader of that slot
    }
    const char* sb = smem + ((ABL & 8) ? 0 : (CS >= 0 ? CS : st_cur)) * STAGE;
    UD_ATT_STAMP(1);                                       // [0] DMA issue

    if constexpr (QB > 1) {
      // ================= QB query blocks per wave (MODE 1): every K / V^T fragment read from LDS feeds QB MFMAs =================
      f32x16 sq[QB][2];
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) sq[qb][kb][r] = -m_q[qb];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const char* kp = sb + (kb * 32 + ql) * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const half8 kf = *(const half8*)(kp + (((ks * 2 + hh) ^ kswz) << 4));
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) sq[qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[qb][ks], sq[qb][kb], 0, 0, 0);
        }
      }
      if (kt == nt - 1 && (p.Nk & (KT - 1))) {           // key tail (last tile only)
        const int kbase = kt * KT + 4 * hh;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kbase + kb * 32 + (r & 3) + 8 * (r >> 2);
            if (key >= p.Nk) {
#pragma unroll
              for (int qb = 0; qb < QB; ++qb) sq[qb][kb][r] = -1.0e30f;
            }
          }
      }
      // per block: how far the tile's row maximum exceeds the running offset (scores are S - m already); deferred like the one-block form
      float mtq[QB];
      float mmax = -1.0e30f;
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        float mt = sq[qb][0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; r += 2) mt = fmaxf(fmaxf(mt, sq[qb][kb][r]), sq[qb][kb][r + 1]);
        mtq[qb] = fmaxf(mt, __shfl_xor(mt, 32, 64));
        mmax = fmaxf(mmax, mtq[qb]);
      }
      if (kt == kt0 || __any(mmax > defer_thr)) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          const float d = kt == kt0 ? mtq[qb] : fmaxf(mtq[qb], 0.0f);
          m_q[qb] += d;
          if (kt != kt0) {                                 // first tile: O and l are still zero (and exp2(-d) may overflow)
            const float alpha = __builtin_amdgcn_exp2f(-d);
            l_q[qb] *= alpha;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
              for (int r = 0; r < 16; ++r) oq[qb][db][r] *= alpha;
          }
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) sq[qb][kb][r] -= d;
        }
      }
      half8 pfq[QB][2][2];
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        float ls = 0.0f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
              f32x2 pv;
              pv[0] = __builtin_amdgcn_exp2f(sq[qb][kb][t * 8 + e]);
              pv[1] = __builtin_amdgcn_exp2f(sq[qb][kb][t * 8 + e + 1]);
              ls += pv[0] + pv[1];
              const half2v ph = __builtin_convertvector(pv, half2v);
              pfq[qb][kb][t][e] = ph[0];
              pfq[qb][kb][t][e + 1] = ph[1];
            }
        l_q[qb] += ls;
      }
      // ---- O^T += V^T P^T, every V^T fragment for all blocks
      const char* vs = sb + KS_BYTES;
      if constexpr (!UD_ATTN_NOPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const char* vrow = vs + (db * 32 + ql) * 128;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const half8 vf = *(const half8*)(vrow + ((((kb * 2 + t) * 2 + hh) ^ kswz) << 4));
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) oq[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pfq[qb][kb][t], oq[qb][db], 0, 0, 0);
          }
      }
      if constexpr (!UD_ATTN_NOPRIO) __builtin_amdgcn_s_setprio(0);
    } else {
    // ---- S^T = K Q^T  (two 32-key blocks), MODE >= 1: minus the running offset m_i
    f32x16 s[2];
    auto scores = [&]() {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = MODE ? -m_i : 0.0f;
        const char* kp = sb + (kb * 32 + ql) * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const half8 kf = *(const half8*)(kp + (((ks * 2 + hh) ^ kswz) << 4));
          if constexpr (ABL & 4) {
            asm volatile("" ::"v"(kf));
            s[kb][ks] += (float)kf[0];
          } else {
            s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[0][ks], s[kb], 0, 0, 0);
          }
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
    };
    constexpr bool NOMAX = UD_ATTN_NOMAX && MODE == 1 && !UD_ATTN_SPLIT_SM;
    if constexpr (!NOMAX) scores();
    // ---- online softmax (lane-local + partner lane ^ 32), max update deferred until it grows by > 2^8 (fp16 P has
    //      constant relative precision, so P up to 256 costs no accuracy; accumulation is fp32)
    auto rowmax = [&]() {
      float mt = s[0][0];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) mt = fmaxf(fmaxf(mt, s[kb][r]), s[kb][r + 1]);
      return fmaxf(mt, __shfl_xor(mt, 32, 64));
    };
#if UD_ATTN_TRACE
    asm volatile("" : "+v"(s[0]), "+v"(s[1]));              // the scores are complete (MFMA results read) before the stamp
    UD_ATT_STAMP(2);                                       // [1] K fragment reads + Q K^T MFMAs
#endif
    float mt = 0.0f;
    if constexpr (!NOMAX) mt = rowmax();
    float ls = 0.0f;
    half8 pf[2][2];
    if constexpr (NOMAX) {
      // optimistic tile: P = exp2(S - m_i) against the offset as it stands; the tile is redone with its exact maximum (Q K^T again from the
      // staged K tile, rescale of O and l) only if a lane's partial row sum says a P may have left the fp16 range -- or on the first tile,
      // where m_i is not yet a maximum of anything.  Straight-line fast path: the redo is a forward branch that rejoins before P V.
      auto exps = [&]() {
        ls = 0.0f;
#if UD_ATTN_MFMASUM
        // row sums on the matrix pipe: v_mfma_f32_4x4x4_16b_f16 with A = ones gives every lane the sum of ITS OWN four B halves (16 blocks
        // of 4 lanes; D[b][i][j] = sum_k A[b][i][k] B[b][k][j]) -- 8 two-pass MFMAs per tile instead of 32 v_add_f32, and the sum is over
        // the fp16-ROUNDED probabilities, i.e. exactly what P V multiplies
        f32x4 la[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        const half4 ones4 = {(half_t)1.0f, (half_t)1.0f, (half_t)1.0f, (half_t)1.0f};
#endif
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
              f32x2 pv;
              pv[0] = __builtin_amdgcn_exp2f(s[kb][t * 8 + e]);
              pv[1] = __builtin_amdgcn_exp2f(s[kb][t * 8 + e + 1]);
#if !UD_ATTN_MFMASUM
              ls += pv[0] + pv[1];
#endif
              const half2v ph = __builtin_convertvector(pv, half2v);
              pf[kb][t][e] = ph[0];
              pf[kb][t][e + 1] = ph[1];
            }
#if UD_ATTN_MFMASUM
            const half4 lo = __builtin_shufflevector(pf[kb][t], pf[kb][t], 0, 1, 2, 3), hi = __builtin_shufflevector(pf[kb][t], pf[kb][t], 4, 5, 6, 7);
            la[0] = __builtin_amdgcn_mfma_f32_4x4x4f16(ones4, lo, la[0], 0, 0, 0);
            la[1] = __builtin_amdgcn_mfma_f32_4x4x4f16(ones4, hi, la[1], 0, 0, 0);
#endif
          }
#if UD_ATTN_MFMASUM
        ls = la[0][0] + la[1][0];
#endif
      };
      scores();
      exps();
      if (__builtin_expect(kt == kt0 || __any(!(ls <= 32768.0f)), 0)) {      // NaN / inf safe: anything not provably small redoes the tile
        scores();
        const float mx = rowmax();
        const float d = kt == kt0 ? mx : fmaxf(mx, 0.0f);
        m_i += d;
        if (kt != kt0) {
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
        exps();
      }
    } else
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
            if constexpr (ABL & 1) {
              pv[0] = s[kb][t * 8 + e];
              pv[1] = s[kb][t * 8 + e + 1];
            } else {
              pv[0] = __builtin_amdgcn_exp2f(fmaf(s[kb][t * 8 + e], c, -mc));
              pv[1] = __builtin_amdgcn_exp2f(fmaf(s[kb][t * 8 + e + 1], c, -mc));
            }
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
      const half2v ones = {(half_t)1.0f, (half_t)1.0f};
#if UD_ATTN_SPLIT_SM
      // The tile in two key-block halves: exp / pack / sum of block 0, then the four P V MFMAs of block 0 with the exp / pack / sum of
      // block 1 placed in their shadows (an MFMA occupies the issue port for one pass of its eight; the wave's own independent VALU
      // work can issue behind it), then the four MFMAs of block 1.  Same arithmetic per element; the row sum is formed per block.
      float lsb[2] = {0.0f, 0.0f};
      auto sm_half = [&](auto KB) {
        constexpr int kb = decltype(KB)::value;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            f32x2 pv;
            pv[0] = __builtin_amdgcn_exp2f(s[kb][t * 8 + e]);
            pv[1] = __builtin_amdgcn_exp2f(s[kb][t * 8 + e + 1]);
            const half2v ph = __builtin_convertvector(pv, half2v);
            if constexpr (MODE == 2) lsb[kb] = __builtin_amdgcn_fdot2(ph, ones, lsb[kb], false);
            else lsb[kb] += pv[0] + pv[1];
            pf[kb][t][e] = ph[0];
            pf[kb][t][e + 1] = ph[1];
          }
      };
      const char* vs2 = sb + KS_BYTES;
      auto pv_half = [&](auto KB) {
        constexpr int kb = decltype(KB)::value;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            const half8 vf = *(const half8*)(vs2 + (db * 32 + ql) * 128 + ((((kb * 2 + t) * 2 + hh) ^ kswz) << 4));
            o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kb][t], o[db], 0, 0, 0);
          }
      };
      sm_half(std::integral_constant<int, 0>{});
      __builtin_amdgcn_sched_barrier(0);
      pv_half(std::integral_constant<int, 0>{});
      sm_half(std::integral_constant<int, 1>{});
      // block-0 MFMAs interleaved with block 1's softmax: per MFMA its V^T fragment read, 4 transcendentals, 6 plain VALU (2 pack + 4 adds)
      // (V^T fragment reads run one group ahead of their MFMA: two up front, then one per group)
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x400, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
        if (g < 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      pv_half(std::integral_constant<int, 1>{});
      l_i += lsb[0] + lsb[1];
    }
#else
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
            if constexpr (MODE == 2) ls = __builtin_amdgcn_fdot2(ph, ones, ls, false);
            else ls += pv[0] + pv[1];
            pf[kb][t][e] = ph[0];
            pf[kb][t][e + 1] = ph[1];
          }
    }
#endif
    l_i += ls;
#if UD_ATTN_TRACE
    asm volatile("" : "+v"(pf[0][0]), "+v"(pf[0][1]), "+v"(pf[1][0]), "+v"(pf[1][1]), "+v"(l_i));
    UD_ATT_STAMP(3);                                       // [2] softmax VALU (max, exchange, exp, sum, pack)
#endif

    // ---- O^T += V^T P^T
    const char* vs = sb + KS_BYTES;
#if UD_ATTN_SPLIT_SM
    if constexpr (MODE == 0)
#endif
    {
    if constexpr (!UD_ATTN_NOPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const char* vrow = vs + (db * 32 + ql) * 128;      // (row >> 1) & 7 == kswz for row = db*32 + ql
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const half8 vf = *(const half8*)(vrow + ((((kb * 2 + t) * 2 + hh) ^ kswz) << 4));
          if constexpr (ABL & 2) {
            asm volatile("" ::"v"(vf), "v"(pf[kb][t]));
            o[db][kb * 2 + t] += (float)vf[0];
          } else {
#if UD_ATTN_OAGPR
            // s_nop 1: the packed P registers may have been written by the VALU instruction just before (VALU write -> MFMA SrcA/B hazard)
            asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(o[db]) : "v"(vf), "v"(pf[kb][t]));
#else
            o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kb][t], o[db], 0, 0, 0);
#endif
          }
        }
    }
    if constexpr (!UD_ATTN_NOPRIO) __builtin_amdgcn_s_setprio(0);
    }

    }   // QB == 1

#if UD_ATTN_TRACE
    UD_ATT_STAMP(4);                                       // [3] V^T fragment reads + P V MFMAs issued (the last MFMAs may still run)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    UD_ATT_STAMP(5);                                       // [4] wait for this wave's DMA pieces of the next tile
    __builtin_amdgcn_s_barrier();
    UD_ATT_STAMP(6);                                       // [5] barrier
#else
    if constexpr (NST == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of tile kt+1 landed; the barrier covers the others'
      if constexpr (!(ABL & 16)) __syncthreads();
    } else {
      // deeper ring: tile kt+1 was issued NST-2 tiles ago; leave the younger tiles' DMAs (2 * PW instructions each) in flight.  Raw barrier:
      // __syncthreads() would make the compiler drain vmcnt to 0 (an LDS-DMA is a pending LDS write to its fence).
      if (ahead) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((NST - 2) * 2 * PW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if constexpr (!(ABL & 16)) __builtin_amdgcn_s_barrier();
    }
#endif
    st_cur = st_cur + 1 == NST ? 0 : st_cur + 1;
    st_new = st_new + 1 == NST ? 0 : st_new + 1;
  };
#if UD_ATTN_UNROLL2
  static_assert(NST == 2, "UD_ATTN_UNROLL2 needs the 2-deep ring");
  for (int kt = kt0; kt < kt1; kt += 2) {
    tile(kt, std::integral_constant<int, 0>{});
    if (kt + 1 < kt1) tile(kt + 1, std::integral_constant<int, 1>{});
  }
#else
  for (int kt = kt0; kt < kt1; ++kt) tile(kt, std::integral_constant<int, -1>{});
#endif

#if UD_ATTN_TRACE
  if (ud_attn_trace_ptr && lane == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) atomicAdd(ud_attn_trace_ptr + i, tsum[i]);
    atomicAdd(ud_attn_trace_ptr + 6, (unsigned long long)(kt1 - kt0));
  }
#endif
  if constexpr (QB > 1) {
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const float lt = l_q[qb] + __shfl_xor(l_q[qb], 32, 64);
      const float inv = 1.0f / lt;
      const int qr = q0 + qb * 32 + ql;
      if (qr < p.Nq) {
        half_t* op = (half_t*)p.O + ((size_t)img * p.q_rows_per_img + qr) * p.ldo + head * 64 + 4 * hh;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            half4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = (half_t)(oq[qb][db][g * 4 + e] * inv);
            *(half4*)(op + db * 32 + g * 8) = h;
          }
      }
    }
    return;
  }
  const float l_tot = l_i + __shfl_xor(l_i, 32, 64);
  const int qr = q0 + ql;
  if constexpr (SPLIT) {
    // ---- split-key mode: un-normalised O^T, the running maximum in natural-log units (MODE 0: m_i is a raw score, P = exp(scale (s - m)))
    //      and the row sum go to part[img][chunk][head * Nq + q][0..65]
    if (qr < p.Nq) {
      float* pp = p.part + ((((size_t)img * nc + chunk) * p.H + head) * p.Nq + qr) * UD_ATTN_PART_LD;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = o[db][g * 4 + e];
          *(f32x4*)(pp + db * 32 + g * 8 + 4 * hh) = v;
        }
      if (hh == 0) {
        pp[64] = MODE ? m_i * 0.6931471805599453f : m_i * p.scale;
        pp[65] = l_tot;
      }
    }
    return;
  }
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

// split-key mode, second step: one 64-lane block per (image, head, query) row
__global__ __launch_bounds__(64) void attention_merge_kernel(const float* part, const float* bias, float* out, int B, int NC, int H, int Nq) {
  const int row = blockIdx.x;                       // (img * H + head) * Nq + q
  const int q = row % Nq, ih = row / Nq;
  const int head = ih % H, img = ih / H;
  const int d = threadIdx.x;
  const size_t cs = (size_t)H * Nq * UD_ATTN_PART_LD;
  const float* pb = part + (size_t)img * NC * cs + ((size_t)head * Nq + q) * UD_ATTN_PART_LD;
  float M = -__builtin_inff();
  for (int c = 0; c < NC; ++c) M = fmaxf(M, pb[c * cs + 64]);
  float L = 0.f, acc = 0.f;
  for (int c = 0; c < NC; ++c) {
    const float e = __expf(pb[c * cs + 64] - M);
    L = fmaf(pb[c * cs + 65], e, L);
    acc = fmaf(pb[c * cs + d], e, acc);
  }
  out[(((size_t)head * B + img) * Nq + q) * 64 + d] = acc / L + (bias ? bias[head * 64 + d] : 0.f);
}

}  // namespace

extern "C" int ud_attention_merge_f32(const float* part, const float* bias, float* out, int B, int NC, int H, int Nq, void* stream) {
  if (!part || !out || B <= 0 || NC <= 0 || H <= 0 || Nq <= 0) {
    ud_set_error("ud_attention_merge_f32: bad argument");
    return UD_ERR_BAD_ARG;
  }
  hipLaunchKernelGGL(attention_merge_kernel, dim3((unsigned)B * H * Nq), dim3(64), 0, (hipStream_t)stream, part, bias, out, B, NC, H, Nq);
  UD_CHECK_LAUNCH("ud_attention_merge_f32 launch");
  return UD_OK;
}

extern "C" int ud_attention_f16(const UdAttention* desc, void* stream) {
  const UdAttention& d = *desc;
  const bool split = d.k_chunk > 0;
  if (split && (!d.part || (d.k_chunk & 63) || d.q_prescaled || d.kv_broadcast)) {
    ud_set_error("ud_attention_f16: split-key mode needs part, k_chunk % 64 == 0, q_prescaled == 0, kv_broadcast == 0");
    return UD_ERR_BAD_ARG;
  }
  if (!d.Q || !d.K || !d.Vt || (!d.O && !split) || d.B <= 0 || d.H <= 0 || d.Nq <= 0 || d.Nk <= 0 || (d.ldq & 7) || (d.ldk & 7) ||
      (d.ldo & 3) || (d.kv_ld & 63) || d.kv_ld < ((d.Nk + 63) & ~63)) {
    ud_set_error("ud_attention_f16: bad argument (ldq/ldk % 8, kv_ld % 64, kv_ld >= roundup(Nk, 64))");
    return UD_ERR_BAD_ARG;
  }
#ifndef UD_ATTN_NW
#define UD_ATTN_NW 4
#endif
  // -DUD_ATTN_NW=8: 8-wave workgroups (256 query rows share each staged K / V^T tile) for long query sequences.  Measured on the encoder
  // shape (B = 8, H = 16, N = 1370), interleaved against the 4-wave build on one box: 90.6 / 90.7 us against 86.3 / 90.4 us -- halving the
  // K / V staging traffic buys nothing (105 instead of 124 VGPRs, same 16 waves per CU), and 6 tiles of 256 rows waste 5 of 48 wave slots
  // per (image, head) where 11 tiles of 128 rows waste 1 of 44.  The product builds the 4-wave form.
  const bool wide = UD_ATTN_NW == 8 && d.Nq >= 512 && !split;
  // UD_ATTN_QB query blocks per wave on the pre-scaled (encoder / decoder MODE 1) path when the query sequence is long enough to fill the chip
  const bool multi = UD_ATTN_QB > 1 && d.q_prescaled && !split && !wide && d.Nq >= 512;
  const int rows_wg = (wide ? 256 : 128) * (multi ? UD_ATTN_QB : 1);
  const int qt = (d.Nq + rows_wg - 1) / rows_wg, pairs = d.B * d.H;
  const int ntile = (d.Nk + 63) / 64;
  const int nc = split ? (ntile + d.k_chunk / 64 - 1) / (d.k_chunk / 64) : 1;
  dim3 grid(8 * ((pairs + 7) / 8) * qt * nc);
  const float thr = (ud_debug_flags_host() & 1) ? -1.0f : 8.0f;
  const int extra_lds = ((ud_debug_flags_host() >> 16) & 255) * 1024;   // tools only: occupancy experiments
  if (split) {
    hipLaunchKernelGGL((attention_kernel<0, 0, 4, true>), grid, dim3(256), 0, (hipStream_t)stream, d, thr);
    UD_CHECK_LAUNCH("ud_attention_f16 (split-key) launch");
    return UD_OK;
  }
#ifdef UD_ABLATE
  switch ((ud_debug_flags_host() >> 8) & 31) {
#define UD_ABL_CASE(X) case X: hipLaunchKernelGGL((attention_kernel<X, 0>), grid, dim3(256), extra_lds, (hipStream_t)stream, d, thr); break;
    UD_ABL_CASE(1) UD_ABL_CASE(2) UD_ABL_CASE(3) UD_ABL_CASE(4) UD_ABL_CASE(7) UD_ABL_CASE(8) UD_ABL_CASE(9) UD_ABL_CASE(24) UD_ABL_CASE(25) UD_ABL_CASE(31)
    default: hipLaunchKernelGGL((attention_kernel<0, 0>), grid, dim3(256), extra_lds, (hipStream_t)stream, d, thr);
  }
#else
#ifndef UD_ATTN_PRE_MODE
#define UD_ATTN_PRE_MODE 1
#endif
  if (wide) {
    if (d.q_prescaled) hipLaunchKernelGGL((attention_kernel<0, UD_ATTN_PRE_MODE, 8>), grid, dim3(512), extra_lds, (hipStream_t)stream, d, thr);
    else hipLaunchKernelGGL((attention_kernel<0, 0, 8>), grid, dim3(512), extra_lds, (hipStream_t)stream, d, thr);
  } else if (multi) {
#if UD_ATTN_QB > 1
    hipLaunchKernelGGL((attention_kernel<0, 1, 4, false, UD_ATTN_QB>), grid, dim3(256), extra_lds, (hipStream_t)stream, d, thr);
#endif
  } else if (d.q_prescaled) hipLaunchKernelGGL((attention_kernel<0, UD_ATTN_PRE_MODE>), grid, dim3(256), extra_lds, (hipStream_t)stream, d, thr);
  else hipLaunchKernelGGL((attention_kernel<0, 0>), grid, dim3(256), extra_lds, (hipStream_t)stream, d, thr);
#endif
  UD_CHECK_LAUNCH("ud_attention_f16 launch");
  return UD_OK;
}
