// fp16 MFMA GEMM family for gfx950:  C[M,N] = A[M,K] * W[N,K]^T  (+ fused epilogues, + implicit-GEMM 3x3 conv A-gather).
//
// Design (MI355X-first, see DESIGN.md section "GEMM"):
//  * 128 x BN x 64 tiles, 4 waves (wave64), v_mfma_f32_16x16x32_f16, fp32 accumulators in registers.
//  * both operands are K-contiguous and go HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR round trip);
//    the LDS image is lane-linear [row][64 halves]; bank conflicts of the ds_read_b128 fragment reads are removed
//    by an XOR swizzle of the 16-byte chunk index, applied on the *global source* address and on the read address:
//    chunk' = chunk ^ ((row >> 1) & 7)  -> the 16 lanes of every ds_read_b128 service group hit 16 distinct slots.
//  * 2-stage LDS ring: tile kt+1 streams in while tile kt is multiplied; one s_barrier per K-tile.
//  * operands are issued as mfma(W_frag, A_frag): the accumulator then holds C^T fragments, i.e. each lane owns
//    4 consecutive n for one m -> 8-byte fp16 / 16-byte fp32 row-major stores.  Tiles that produce V^T for the
//    attention kernel use the opposite order (4 consecutive tokens per lane -> 8-byte stores into [head][d][token]).
//  * workgroup -> tile map is XCD-aware: the 8 XCDs get contiguous ranges of the tile list (private L2 reuse).
#include "ud_common.h"

namespace {

template <int BN_, int WM_, int WN_>
struct Cfg {
  static constexpr int BM = 128, BN = BN_, BK = 64, WM = WM_, WN = WN_;
  static constexpr int NWM = BM / WM, NWN = BN / WN;
  static constexpr int TM = WM / 16, TN = WN / 16;
  static constexpr int A_BYTES = BM * BK * 2;
  static constexpr int B_BYTES = BN * BK * 2;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int A_INSTR = BM / 32;  // glds per wave per tile (8 rows each)
  static constexpr int B_INSTR = BN / 32;
  static_assert(NWM * NWN == 4, "4 waves");
};

struct ConvLane {
  long long base;  // element offset of the image
  int y, x;
  int valid;
};

template <class C, int EPI, int AMODE, bool SWAP>
__device__ __forceinline__ void gemm_body(const UdGemm& p, char* smem, int m0, int n0, const half_t* A, const half_t* W,
                                          const float* bias, char* out, char* out2, const float* w2, float b2, float post_add) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv / C::NWN, wn = wv % C::NWN;
  const int nk = p.K >> 6;

  // ---------------- loader geometry: lane -> (row within 8-row group, chunk position) ----------------
  const int lrow = lane >> 3;   // 0..7
  const int cpos = lane & 7;    // chunk slot inside the 128-byte LDS row
  const half_t* aptr[C::A_INSTR];
  const half_t* bptr[C::B_INSTR];
  int a_csrc[C::A_INSTR];
  ConvLane cl[C::A_INSTR];
  const float inv_cc = (AMODE != UD_A_DENSE) ? 1.0f / (float)(p.Cin >> 3) : 0.0f;
#pragma unroll
  for (int i = 0; i < C::A_INSTR; ++i) {
    const int r = (wv * C::A_INSTR + i) * 8 + lrow;
    const int cs = cpos ^ ((r >> 1) & 7);
    a_csrc[i] = cs;
    int m = m0 + r;
    m = m < p.M ? m : p.M - 1;
    if constexpr (AMODE == UD_A_DENSE) {
      aptr[i] = A + (size_t)m * p.lda + cs * 8;
    } else {
      const int img = m / p.rows_img;
      const int pp = m - img * p.rows_img;
      const int y = pp / p.Wimg;
      cl[i].base = (long long)img * p.img_stride;
      cl[i].y = y;
      cl[i].x = pp - y * p.Wimg;
      cl[i].valid = pp < p.Himg * p.Wimg;
    }
  }
#pragma unroll
  for (int i = 0; i < C::B_INSTR; ++i) {
    const int r = (wv * C::B_INSTR + i) * 8 + lrow;
    const int cs = cpos ^ ((r >> 1) & 7);
    int n = n0 + r;
    n = n < p.N ? n : p.N - 1;
    bptr[i] = W + (size_t)n * p.ldw + cs * 8;
  }

  auto issue = [&](int kt, int stage) {
    char* sb = smem + stage * C::STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < C::A_INSTR; ++i) {
      const half_t* src;
      if constexpr (AMODE == UD_A_DENSE) {
        src = aptr[i] + kt * 64;
      } else {
        const int kc = kt * 8 + a_csrc[i];                       // 8-channel chunk index along K = (tap, cin)
        const int tap = (int)(((float)kc + 0.5f) * inv_cc);
        const int cch = (kc - tap * (p.Cin >> 3)) << 3;
        const int t3 = (tap * 11) >> 5;                           // tap / 3 for tap < 12
        int yy = cl[i].y + t3 - 1;
        int xx = cl[i].x + (tap - t3 * 3) - 1;
        bool ok = cl[i].valid && tap < 9;
        if constexpr (AMODE == UD_A_CONV3_ZERO) {
          ok = ok && (unsigned)yy < (unsigned)p.Himg && (unsigned)xx < (unsigned)p.Wimg;
        } else {
          yy = yy < 0 ? -yy : (yy >= p.Himg ? 2 * p.Himg - 2 - yy : yy);
          xx = xx < 0 ? -xx : (xx >= p.Wimg ? 2 * p.Wimg - 2 - xx : xx);
        }
        src = ok ? A + cl[i].base + ((long long)(yy * p.Wimg + xx)) * p.cstride + p.coff + cch : (const half_t*)p.zeros;
      }
      ud_glds16(src, sb + (wv * C::A_INSTR + i) * 1024);
    }
#pragma unroll
    for (int i = 0; i < C::B_INSTR; ++i) ud_glds16(bptr[i] + kt * 64, sb + C::A_BYTES + (wv * C::B_INSTR + i) * 1024);
  };

  // ---------------- fragment read geometry ----------------
  const int frow = lane & 15;
  const int fq = lane >> 4;              // k-chunk within a 32-wide k-step
  const int fswz = frow >> 1;            // (row >> 1) & 7 for row = 16*t + frow
  const int a_row_off = (wm * C::WM + frow) * 128;
  const int b_row_off = C::A_BYTES + (wn * C::WN + frow) * 128;

  f32x4 acc[C::TM][C::TN];
#pragma unroll
  for (int i = 0; i < C::TM; ++i)
#pragma unroll
    for (int j = 0; j < C::TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  issue(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
    const char* sb = smem + (kt & 1) * C::STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = ((ks * 4 + fq) ^ fswz) << 4;
      half8 af[C::TM], bf[C::TN];
#pragma unroll
      for (int i = 0; i < C::TM; ++i) af[i] = *(const half8*)(sb + a_row_off + i * 2048 + coff);
#pragma unroll
      for (int j = 0; j < C::TN; ++j) bf[j] = *(const half8*)(sb + b_row_off + j * 2048 + coff);
#pragma unroll
      for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j) {
          if constexpr (SWAP)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
          else
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
  }

  // ---------------- epilogue ----------------
  if constexpr (!SWAP) {
    // V^T tiles of UD_EPI_QKV: lane owns tokens mb..mb+3 (one image: tok_per_img % 4 == 0) for column n.
    static_assert(EPI == UD_EPI_QKV, "non-swapped orientation only for V^T");
    half_t* vt = (half_t*)out2;
#pragma unroll
    for (int i = 0; i < C::TM; ++i) {
      const int mb = m0 + wm * C::WM + i * 16 + 4 * (lane >> 4);
      if (mb >= p.M) continue;
      const int img = mb / p.tok_per_img;
      const int t = mb - img * p.tok_per_img;
#pragma unroll
      for (int j = 0; j < C::TN; ++j) {
        const int n = n0 + wn * C::WN + j * 16 + (lane & 15);
        if (n >= p.N) continue;
        const float bv = bias ? bias[n] : 0.0f;
        const int nv = n - p.vsplit;
        const int hd = nv >> 6, d = nv & 63;
        half4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = (half_t)(acc[i][j][r] + bv);
        *(half4*)(vt + (((size_t)img * p.heads_v + hd) * 64 + d) * p.kv_ld + t) = h;
      }
    }
    return;
  } else {
#pragma unroll
    for (int i = 0; i < C::TM; ++i) {
      const int m = m0 + wm * C::WM + i * 16 + (lane & 15);
      const bool mok = m < p.M;
      // output row remap
      int orow = m, arow = m;
      if (p.rows_in > 0) {
        const int img = m / p.rows_in;
        const int pr = m - img * p.rows_in;
        orow = img * p.rows_out + pr + p.row_off;
        arow = pr + p.add_row_off;
      }
      if constexpr (EPI == UD_EPI_HEAD) {
        float part = 0.0f;
#pragma unroll
        for (int j = 0; j < C::TN; ++j) {
          const int nb = n0 + wn * C::WN + j * 16 + 4 * (lane >> 4);
          if (nb < p.N) {
            const f32x4 bv = *(const f32x4*)(bias + nb);
            const f32x4 wv2 = *(const f32x4*)(w2 + nb);
#pragma unroll
            for (int r = 0; r < 4; ++r) part += ud_lrelu(acc[i][j][r] + bv[r]) * wv2[r];
          }
        }
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        if (mok && (lane >> 4) == 0) {
          float y = part + b2;
          y = fminf(fmaxf(y, -8.0f), 8.0f);
          ((float*)out)[orow] = __expf(y + post_add);
        }
        continue;
      }
      if constexpr (EPI == UD_EPI_D2S) {
        // m -> (img, y, x) on the input grid
        const int img = m / p.d2s_rows_in_img;
        const int pp = m - img * p.d2s_rows_in_img;
        const int y = pp / p.d2s_Win, x = pp - y * p.d2s_Win;
        const bool ok = mok && pp < p.d2s_Hin * p.d2s_Win;
        const int k = p.d2s_k;
        const int Wout = p.d2s_Win * k;
#pragma unroll
        for (int j = 0; j < C::TN; ++j) {
          const int nb = n0 + wn * C::WN + j * 16 + 4 * (lane >> 4);
          if (!ok || nb >= p.N) continue;
          const int ac = nb / p.d2s_Co;
          const int o = nb - ac * p.d2s_Co;
          const int a = ac / k, c = ac - a * k;
          const long long pix = (long long)img * p.d2s_out_img_pix + (long long)(y * k + a) * Wout + (x * k + c);
          float* dst = (float*)out + pix * p.ldc + o;
          f32x4 v = *(f32x4*)dst;
          const f32x4 bv = *(const f32x4*)(bias + o);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += acc[i][j][r] + bv[r];
          *(f32x4*)dst = v;
          if (out2) {
            half4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = (half_t)ud_act(v[r], p.act2);
            *(half4*)((half_t*)out2 + pix * p.ldc2 + o) = h;
          }
        }
        continue;
      }
#pragma unroll
      for (int j = 0; j < C::TN; ++j) {
        const int nb = n0 + wn * C::WN + j * 16 + 4 * (lane >> 4);
        if (!mok || nb >= p.N) continue;
        f32x4 v = acc[i][j];
        if (bias) {
          const f32x4 bv = *(const f32x4*)(bias + nb);
          v += bv;
        }
        if (p.add) {
          const f32x4 av = *(const f32x4*)(p.add + (size_t)arow * p.ldadd + nb);
          v += av;
        }
        if constexpr (EPI == UD_EPI_F16 || EPI == UD_EPI_QKV) {
          half4 h;
#pragma unroll
          for (int r = 0; r < 4; ++r) h[r] = (half_t)ud_act(v[r], p.act);
          *(half4*)((half_t*)out + (size_t)orow * p.ldc + nb) = h;
        } else if constexpr (EPI == UD_EPI_F32) {
          float* dst = (float*)out + (size_t)orow * p.ldc + nb;
          if (p.accumulate) {
            const f32x4 old = *(const f32x4*)dst;
            v += old;
          }
          *(f32x4*)dst = v;
          if (out2) {
            half4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = (half_t)ud_act(v[r], p.act2);
            *(half4*)((half_t*)out2 + (size_t)orow * p.ldc2 + nb) = h;
          }
        }
      }
    }
  }
}

template <class C, int EPI, int AMODE>
__global__ __launch_bounds__(256) void gemm_kernel(const UdGemm p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_n = (p.N + C::BN - 1) / C::BN;
  const int tiles_m = (p.M + C::BM - 1) / C::BM;
  const int nblk = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {  // bijective XCD-aware remap: block b runs on XCD b % 8 -> give each XCD a contiguous range of tiles
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = bid / tiles_n;
  const int tile_n = bid - tile_m * tiles_n;
  const int m0 = tile_m * C::BM, n0 = tile_n * C::BN;
  const int g = blockIdx.z;
  const half_t* A = (const half_t*)p.A + (long long)g * p.gA;
  const half_t* W = (const half_t*)p.W + (long long)g * p.gW;
  const float* bias = p.bias ? p.bias + (long long)g * p.gBias : nullptr;
  char* out = (char*)p.out + (long long)g * p.gOut * ((EPI == UD_EPI_F16 || EPI == UD_EPI_QKV) ? 2 : 4);
  char* out2 = p.out2 ? (char*)p.out2 + (long long)g * p.gOut2 * 2 : nullptr;
  const float* w2 = p.w2 ? p.w2 + (long long)g * p.gW2 : nullptr;
  const float b2 = g == 0 ? p.b2 : p.b2_g1;
  const float post_add = g == 0 ? p.post_add : p.post_add_g1;
  if constexpr (EPI == UD_EPI_QKV) {
    if (n0 >= p.vsplit) {
      gemm_body<C, EPI, AMODE, false>(p, smem, m0, n0, A, W, bias, out, out2, w2, b2, post_add);
      return;
    }
  }
  gemm_body<C, EPI, AMODE, true>(p, smem, m0, n0, A, W, bias, out, out2, w2, b2, post_add);
}

template <class C, int EPI, int AMODE>
int launch(const UdGemm& d, hipStream_t s) {
  const int tiles_n = (d.N + C::BN - 1) / C::BN;
  const int tiles_m = (d.M + C::BM - 1) / C::BM;
  const int lds = 2 * C::STAGE_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_kernel<C, EPI, AMODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  dim3 grid(tiles_m * tiles_n, 1, d.groups > 0 ? d.groups : 1);
  hipLaunchKernelGGL((gemm_kernel<C, EPI, AMODE>), grid, dim3(256), lds, s, d);
  UD_CHECK_LAUNCH("ud_gemm_f16 launch");
  return UD_OK;
}

template <int EPI, int AMODE>
int dispatch_bn(const UdGemm& d, hipStream_t s) {
  if (d.N > 64) return launch<Cfg<128, 64, 64>, EPI, AMODE>(d, s);
  if (d.N > 32) return launch<Cfg<64, 64, 32>, EPI, AMODE>(d, s);
  return launch<Cfg<32, 32, 32>, EPI, AMODE>(d, s);
}

}  // namespace

extern "C" int ud_gemm_f16(const UdGemm* desc, void* stream) {
  const UdGemm& d = *desc;
  hipStream_t s = (hipStream_t)stream;
  if (!d.A || !d.W || !d.out || d.M <= 0 || d.N <= 0 || d.K <= 0 || (d.K & 63) || (d.N & 3)) {
    ud_set_error("ud_gemm_f16: bad argument (need K % 64 == 0, N % 4 == 0)");
    return UD_ERR_BAD_ARG;
  }
  if (d.amode != UD_A_DENSE && (!d.zeros || (d.Cin & 7) || d.K < 9 * d.Cin || d.rows_img < d.Himg * d.Wimg)) {
    ud_set_error("ud_gemm_f16: bad conv geometry");
    return UD_ERR_BAD_ARG;
  }
  if (d.epi == UD_EPI_QKV) {
    if (d.amode != UD_A_DENSE || !d.out2 || (d.vsplit % 128) || (d.tok_per_img & 3) || (d.kv_ld & 3) || d.N <= 64) {
      ud_set_error("ud_gemm_f16: bad QKV epilogue geometry");
      return UD_ERR_BAD_ARG;
    }
    return launch<Cfg<128, 64, 64>, UD_EPI_QKV, UD_A_DENSE>(d, s);
  }
  if (d.epi == UD_EPI_D2S) {
    if (d.amode != UD_A_DENSE || (d.d2s_Co & 3)) {
      ud_set_error("ud_gemm_f16: bad D2S epilogue geometry");
      return UD_ERR_BAD_ARG;
    }
    return dispatch_bn<UD_EPI_D2S, UD_A_DENSE>(d, s);
  }
  if (d.epi == UD_EPI_HEAD) {
    if (d.amode != UD_A_CONV3_REFLECT || d.N != 32 || !d.w2 || !d.bias) {
      ud_set_error("ud_gemm_f16: HEAD epilogue needs reflect conv, N == 32");
      return UD_ERR_BAD_ARG;
    }
    return launch<Cfg<32, 32, 32>, UD_EPI_HEAD, UD_A_CONV3_REFLECT>(d, s);
  }
  if (d.epi == UD_EPI_F16) {
    if (d.amode == UD_A_DENSE) return dispatch_bn<UD_EPI_F16, UD_A_DENSE>(d, s);
    if (d.amode == UD_A_CONV3_ZERO) return dispatch_bn<UD_EPI_F16, UD_A_CONV3_ZERO>(d, s);
    return dispatch_bn<UD_EPI_F16, UD_A_CONV3_REFLECT>(d, s);
  }
  if (d.epi == UD_EPI_F32) {
    if (d.amode == UD_A_DENSE) return dispatch_bn<UD_EPI_F32, UD_A_DENSE>(d, s);
    if (d.amode == UD_A_CONV3_ZERO) return dispatch_bn<UD_EPI_F32, UD_A_CONV3_ZERO>(d, s);
  }
  ud_set_error("ud_gemm_f16: unsupported epi/amode combination");
  return UD_ERR_UNSUPPORTED;
}
