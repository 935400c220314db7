// ConvNeXt-side kernels of the UniDepthV1 path (reference unidepth/models/backbones/convnext.py:130-223,245-266,370-383,447-458 and
// layers/convnext.py:5-44): everything that is not a GEMM.  All HBM / L2-bound, NHWC (channels contiguous), wave64.
//   dwconv7_kernel        depth-wise 7x7 convolution, zero padding 3, fp32 in / fp32 out (+ bias)
//   ln_patchify2_kernel   LayerNorm2d statistics (affine folded into the following conv) + im2col of the 2x2 stride-2 down-sampling conv
//   patchify4_kernel      im2col of the 4x4 stride-4 stem conv from the NCHW network image
//   max_kernel            running element-wise max over a stage's block outputs (utils/misc.py:18-21 max_stack)
//   spatial_mean_kernel   per-image mean over all pixels ("class tokens" of the ConvNeXt wrapper, convnext.py:458)
#include "ud_common.h"

namespace {

// One wave = 8 consecutive output pixels of one image row x 256 channels (4 per lane, 16-byte accesses).  Per filter row the wave
// loads the 14 input pixels its 8 outputs touch once and feeds each to the (up to 7) outputs it belongs to: 98 loads for 392
// multiply-adds per lane-channel instead of 392.  Weights are stored tap-major [49][C] so a tap is one coalesced 16-byte load.
__global__ __launch_bounds__(256) void dwconv7_kernel(const UdDwConv7 p) {
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int xt = (p.W + 7) >> 3;                       // x tiles per row
  const int tile = blockIdx.x * 4 + wv;
  const int total = p.B * p.H * xt;
  if (tile >= total) return;
  const int x0 = (tile % xt) << 3;
  const int y = (tile / xt) % p.H;
  const int b = tile / (xt * p.H);
  const int c = blockIdx.y * 256 + lane * 4;
  if (c >= p.C) return;
  const float* img = p.x + (size_t)b * p.H * p.W * p.ldx;
  f32x4 acc[8];
  const f32x4 bv = p.bias ? *(const f32x4*)(p.bias + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = bv;
  for (int ky = 0; ky < 7; ++ky) {
    const int iy = y + ky - 3;
    if ((unsigned)iy >= (unsigned)p.H) continue;        // wave-uniform
    f32x4 wr[7];
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) wr[kx] = *(const f32x4*)(p.w + (size_t)(ky * 7 + kx) * p.C + c);
    const float* row = img + (size_t)iy * p.W * p.ldx + c;
#pragma unroll
    for (int j = 0; j < 14; ++j) {
      const int ix = x0 + j - 3;
      if ((unsigned)ix >= (unsigned)p.W) continue;      // wave-uniform
      const f32x4 v = *(const f32x4*)(row + (size_t)ix * p.ldx);
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        const int kx = j - o;
        if (kx >= 0 && kx < 7) acc[o] += v * wr[kx];
      }
    }
  }
  float* out = p.y + (((size_t)b * p.H + y) * p.W + x0) * p.ldy + c;
#pragma unroll
  for (int o = 0; o < 8; ++o)
    if (x0 + o < p.W) *(f32x4*)(out + (size_t)o * p.ldy) = acc[o];
}


// LDS-tiled variant (C % 64 == 0; every ConvNeXt / CvnxtBlock width is): block = 8 x 16 output pixels x 64 channels.  The 14 x 22 x 64
// fp32 halo (77 KB: two blocks per CU) is loaded ONCE per block (16 B per lane, one 256-byte channel run per pixel), then thread
// (channel = tid & 63, wave = row pair) slides along x in registers: 22 LDS reads (conflict-free: the lanes of a wave are 64 consecutive
// channels) feed 112 multiply-adds per filter row.  The per-wave kernel above re-read every input pixel ~12 x through L2 and ran at
// 3.4 TFLOP/s (44 % of the V1 step at bs = 16).
constexpr int DW_TY = 8, DW_TX = 16, DW_HY = DW_TY + 6, DW_HX = DW_TX + 6;
__global__ __launch_bounds__(256, 2) void dwconv7_lds_kernel(const UdDwConv7 p) {
  extern __shared__ __attribute__((aligned(16))) float halo[];          // [DW_HY * DW_HX][64]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int tx_n = (p.W + DW_TX - 1) / DW_TX, ty_n = (p.H + DW_TY - 1) / DW_TY;
  const int t = blockIdx.x;
  const int x0 = (t % tx_n) * DW_TX, y0 = ((t / tx_n) % ty_n) * DW_TY, b = t / (tx_n * ty_n);
  const int cb = blockIdx.y * 64;
  // ---- halo load by LDS-DMA: one wave instruction = 4 consecutive halo pixels x 64 channels = 1 KB, landing at wave-uniform base +
  // lane * 16; pixels outside the image use an offset beyond the descriptor's range and arrive as zeros.  All ~20 loads of a wave are
  // in flight together (the global_load -> VGPR -> ds_write version serialised them on the register round trip: 14 us per block).
  const int sub = lane >> 4, c4 = (lane & 15) * 4;
  const ud_rsrc_t rs = ud_make_rsrc(p.x, (unsigned)((size_t)p.B * p.H * p.W * p.ldx * sizeof(float)));
  for (int i = wv; i < (DW_HY * DW_HX + 3) / 4; i += 4) {
    const int hp = i * 4 + sub;
    const int hy = hp / DW_HX, hx = hp - hy * DW_HX;
    const int iy = y0 + hy - 3, ix = x0 + hx - 3;
    const bool ok = hp < DW_HY * DW_HX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    const unsigned off = ok ? (unsigned)(((((size_t)b * p.H + iy) * p.W + ix) * p.ldx + cb + c4) * sizeof(float)) : 0xfffffff0u;
    ud_bufl16(rs, off, 0, halo + i * 256);
  }
  const int c = cb + lane;
  f32x2 wk[25];                  // taps in register pairs: the packed FMA broadcasts either half (op_sel), no splat copies
#pragma unroll
  for (int k = 0; k < 25; ++k) wk[k] = (f32x2){p.w[(size_t)(2 * k) * p.C + c], k < 24 ? p.w[(size_t)(2 * k + 1) * p.C + c] : 0.f};
  const float bv = p.bias ? p.bias[c] : 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // two neighbouring outputs per packed fp32 FMA (v_pk_fma_f32: 56 per filter row instead of 112 scalar FMAs): output pair
  // (2q, 2q+1) at tap kx needs the input pair starting at column 2q + kx -- an even-aligned pair pe[] for even kx, an odd-aligned
  // one po[] for odd kx; both come straight from LDS (ds_read2st64_b32: two pixels of this lane's channel per read).  Per output
  // the taps are still accumulated ky-major, kx ascending, one fused multiply-add each: the same bits as the scalar loop.
  float keep0[DW_TX], keep1[DW_TX];          // stats_out only: the wave's two output rows, until the halo can be overwritten (two arrays: no dynamic index)
#pragma unroll 1
  for (int rr = 0; rr < 2; ++rr) {
    const int r = wv * 2 + rr;
    f32x2 acc[DW_TX / 2];
#pragma unroll
    for (int o = 0; o < DW_TX / 2; ++o) acc[o] = (f32x2){bv, bv};
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      const float* hrow = halo + (r + ky) * DW_HX * 64 + lane;
      f32x2 pe[DW_HX / 2], po[DW_HX / 2 - 1];
#pragma unroll
      for (int i = 0; i < DW_HX / 2; ++i) pe[i] = (f32x2){hrow[(2 * i) * 64], hrow[(2 * i + 1) * 64]};
#pragma unroll
      for (int i = 0; i < DW_HX / 2 - 1; ++i) po[i] = (f32x2){hrow[(2 * i + 1) * 64], hrow[(2 * i + 2) * 64]};
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        const f32x2 wp = wk[(ky * 7 + kx) >> 1];
#pragma unroll
        for (int q = 0; q < DW_TX / 2; ++q) {
          const f32x2 v = (kx & 1) ? po[q + (kx >> 1)] : pe[q + (kx >> 1)];
          // acc += v * broadcast(wp.lo | wp.hi): the tap is picked with op_sel (the compiler materialises a {w, w} pair per tap)
          if ((ky * 7 + kx) & 1) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc[q]) : "v"(v), "v"(wp));
          else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[q]) : "v"(v), "v"(wp));
        }
      }
    }
    const int y = y0 + r;
    if (y < p.H) {
      if (p.y) {
        float* out = p.y + (((size_t)b * p.H + y) * p.W + x0) * p.ldy + c;
#pragma unroll
        for (int o = 0; o < DW_TX; ++o)
          if (x0 + o < p.W) out[(size_t)o * p.ldy] = acc[o >> 1][o & 1];
      }
      if (p.y16) {                                  // the RAW fp16 row of a LayerNorm-folded consumer (a wave = 64 consecutive channels = 128 B per pixel)
        half_t* out = (half_t*)p.y16 + (((size_t)b * p.H + y) * p.W + x0) * p.ldy16 + c;
#pragma unroll
        for (int o = 0; o < DW_TX; ++o)
          if (x0 + o < p.W) out[(size_t)o * p.ldy16] = (half_t)acc[o >> 1][o & 1];
      }
    }
    if (p.stats_out) {
#pragma unroll
      for (int o = 0; o < DW_TX; ++o) {
        if (rr == 0) keep0[o] = acc[o >> 1][o & 1];
        else keep1[o] = acc[o >> 1][o & 1];
      }
    }
  }
  if (p.stats_out) {
    // per pixel (sum, sum of squares) over this block's 64 channels, from the fp32 values: the tile goes through LDS (the halo is dead: every wave is
    // past its last read after the barrier), [channel][pixel] with a row stride of 129 floats -- lane = channel on the way in, lane = pixel on the way
    // out, both conflict-free -- and thread t < 128 adds pixel t's 64 channels in channel order (a fixed order: reproducible, position-independent).
    __syncthreads();
    float* tile = halo;
#pragma unroll
    for (int o = 0; o < DW_TX; ++o) {
      tile[lane * 129 + (wv * 2) * DW_TX + o] = keep0[o];
      tile[lane * 129 + (wv * 2 + 1) * DW_TX + o] = keep1[o];
    }
    __syncthreads();
    if (tid < DW_TY * DW_TX) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
      for (int ch = 0; ch < 64; ++ch) {
        const float v = tile[ch * 129 + tid];
        s1 += v;
        s2 = __builtin_fmaf(v, v, s2);
      }
      const int yy = y0 + tid / DW_TX, xx = x0 + tid % DW_TX;
      if (yy < p.H && xx < p.W) {
        f32x2 o2;
        o2[0] = s1; o2[1] = s2;
        // system-scope write-through (sc0 sc1): the block that reduces them may sit on another XCD, whose L2 is not coherent with this one's
        float* dst = p.stats_out + ((((size_t)b * p.H + yy) * p.W + xx) * (p.C >> 6) + blockIdx.y) * 2;
        asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_nop 2" ::"v"(dst), "v"(o2) : "memory");
      }
    }
    if (p.stats_final) {
      // the last of the tile's C / 64 channel blocks reduces: partial sums out (vmcnt(0)), one ticket per pixel tile (wraps to zero: no reset launch), the
      // drawer of the last ticket adds the slabs in slab order -- the result depends on nothing but the values (the fence-free exchange of gemm.hip's
      // row_stats_final).  The ticket value travels through LDS (the tile staging area is dead after the sums above).
      const unsigned nslab = (unsigned)(p.C >> 6);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      unsigned* flag = (unsigned*)halo;
      if (tid == 0) *flag = atomicInc(p.stats_ticket + blockIdx.x, nslab - 1u);
      __syncthreads();
      if (*flag == nslab - 1u && tid < DW_TY * DW_TX) {
        const int yy = y0 + tid / DW_TX, xx = x0 + tid % DW_TX;
        if (yy < p.H && xx < p.W) {
          const size_t pix = ((size_t)b * p.H + yy) * p.W + xx;
          const float* src = p.stats_out + pix * nslab * 2;
          // 16 unconditional loads (slabs past the last one re-read slab 0 and are masked out of the sums): all in flight together, no branches
          f32x2 q[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const float* a = src + 2 * ((unsigned)k < nslab ? k : 0);
            asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(q[k]) : "v"(a) : "memory");
          }
          // the loaded registers are operands of the wait: nothing reads them before it
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]), "+v"(q[8]), "+v"(q[9]),
                       "+v"(q[10]), "+v"(q[11]), "+v"(q[12]), "+v"(q[13]), "+v"(q[14]), "+v"(q[15])::"memory");
          float t1 = 0.f, t2 = 0.f;
#pragma unroll
          for (int k = 0; k < 16; ++k)
            if ((unsigned)k < nslab) { t1 += q[k][0]; t2 += q[k][1]; }
          const float inv = 1.0f / (float)p.C;
          const float mean = t1 * inv;
          const float var = fmaxf(__builtin_fmaf(-mean, mean, t2 * inv), 0.0f);
          f32x2 oo;
          oo[0] = rsqrtf(var + p.ln_eps);
          oo[1] = -mean * oo[0];
          *(f32x2*)(p.stats_final + 2 * pix) = oo;
        }
      }
    }
  }
}

// LayerNorm2d (eps) over the C channels of every pixel, statistics only, written as fp16 straight into the im2col image of the
// following Conv2d(k = 2, s = 2, padding 0): pixel (y, x) -> row (b, y / 2, x / 2), columns ((y & 1) * 2 + (x & 1)) * C + c.
// An odd last row / column is dropped, as the convolution drops it.  One wave per pixel, C <= 2048.
template <int NIT>
__global__ __launch_bounds__(256) void ln_patchify2_kernel(const float* x, half_t* out, int B, int H, int W, int C, int ldo, float eps) {
  const int lane = threadIdx.x & 63;
  const int Ho = H >> 1, Wo = W >> 1;
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= (long long)B * Ho * 2 * Wo * 2) return;
  const int xx = (int)(r % (2 * Wo));
  const int yy = (int)((r / (2 * Wo)) % (2 * Ho));
  const int b = (int)(r / ((long long)4 * Wo * Ho));
  const float* src = x + (((size_t)b * H + yy) * W + xx) * C;
  f32x4 v[NIT];
  float s = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = it * 256 + lane * 4;
    if (c < C) {
      v[it] = *(const f32x4*)(src + c);
      s += (v[it][0] + v[it][1]) + (v[it][2] + v[it][3]);
    } else {
      v[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  const float mean = ud_wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = it * 256 + lane * 4;
    if (c < C) {
#pragma unroll
      for (int e = 0; e < 4; ++e) q += (v[it][e] - mean) * (v[it][e] - mean);
    }
  }
  const float rstd = rsqrtf(ud_wave_sum(q) / (float)C + eps);
  half_t* dst = out + (((size_t)b * Ho + (yy >> 1)) * Wo + (xx >> 1)) * ldo + (size_t)((yy & 1) * 2 + (xx & 1)) * C;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = it * 256 + lane * 4;
    if (c < C) {
      half4 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) h[e] = (half_t)((v[it][e] - mean) * rstd);
      *(half4*)(dst + c) = h;
    }
  }
}

// stem im2col: image fp32 NCHW [B,3,H,W] -> fp16 [B*(H/4)*(W/4), ldo], column = c*16 + ky*4 + kx (the order of Conv2d's [Cout,3,4,4] rows)
__global__ __launch_bounds__(256) void patchify4_kernel(const float* img, half_t* out, int B, int H, int W, int ldo) {
  const int Ho = H >> 2, Wo = W >> 2;
  const long long total = (long long)B * Ho * Wo * 48;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int k = (int)(idx % 48);
    const long long m = idx / 48;
    const int xo = (int)(m % Wo);
    const int yo = (int)((m / Wo) % Ho);
    const int b = (int)(m / ((long long)Wo * Ho));
    const int c = k >> 4, ky = (k >> 2) & 3, kx = k & 3;
    out[m * ldo + k] = (half_t)img[(((size_t)b * 3 + c) * H + yo * 4 + ky) * W + xo * 4 + kx];
  }
}

__global__ __launch_bounds__(256) void max_kernel(float* dst, const float* src, long long n4, int init) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const f32x4 s = ((const f32x4*)src)[i];
    if (init) {
      ((f32x4*)dst)[i] = s;
    } else {
      f32x4 d = ((f32x4*)dst)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = fmaxf(d[e], s[e]);
      ((f32x4*)dst)[i] = d;
    }
  }
}

// out[b, c] = mean over the HW pixels of x[b, :, c].  Block = (image, 256-channel chunk); 4 waves split the pixels, LDS combine.
__global__ __launch_bounds__(256) void spatial_mean_kernel(const float* x, float* out, int HW, int C, int ldo) {
  __shared__ f32x4 part[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int b = blockIdx.x;
  const int c = blockIdx.y * 256 + lane * 4;
  f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    const float* src = x + (size_t)b * HW * C + c;
    for (int p = wv; p < HW; p += 4) s += *(const f32x4*)(src + (size_t)p * C);
  }
  part[wv][lane] = s;
  __syncthreads();
  if (wv == 0 && c < C) {
    const f32x4 t = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    *(f32x4*)(out + (size_t)b * ldo + c) = t * (1.0f / (float)HW);
  }
}

}  // namespace

extern "C" int ud_dwconv7_nhwc_f32(const UdDwConv7* desc, void* stream) {
  const UdDwConv7& d = *desc;
  if (!d.x || !d.w || (!d.y && !d.y16) || d.B <= 0 || d.H <= 0 || d.W <= 0 || d.C <= 0 || (d.C & 3) || (d.ldx & 3) || d.ldx < d.C ||
      (d.y && ((d.ldy & 3) || d.ldy < d.C))) {
    ud_set_error("ud_dwconv7_nhwc_f32: bad argument (C, ldx, ldy % 4 == 0)");
    return UD_ERR_BAD_ARG;
  }
  if (d.y16 || d.stats_out) {
    if (!d.y16 || (d.C & 63) || d.ldy16 < d.C || (d.stats_out && (d.C >> 6) > 16) || (d.stats_final && (!d.stats_out || !d.stats_ticket)) || (double)d.B * d.H * d.W * d.ldx * 4.0 >= 4294967000.0) {
      ud_set_error("ud_dwconv7_nhwc_f32: y16 / stats_out need C % 64 == 0 (C <= 1024 for stats_out), ldy16 >= C and an image the LDS-tiled kernel addresses");
      return UD_ERR_UNSUPPORTED;
    }
  }
  // the LDS-tiled kernel addresses the image through one buffer descriptor (32-bit byte offsets)
  if ((d.C & 63) == 0 && (double)d.B * d.H * d.W * d.ldx * 4.0 < 4294967000.0) {
    constexpr int lds = DW_HY * DW_HX * 64 * 4;
    static bool attr_set[UD_MAX_DEVICES];
    if (!ud_attr_once(attr_set)) {
      if (hipFuncSetAttribute((const void*)dwconv7_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
        ud_set_error("ud_dwconv7_nhwc_f32: cannot reserve the halo tile in LDS");
        return UD_ERR_LAUNCH;
      }
    }
    const long long tiles = (long long)d.B * ((d.H + DW_TY - 1) / DW_TY) * ((d.W + DW_TX - 1) / DW_TX);
    hipLaunchKernelGGL(dwconv7_lds_kernel, dim3((unsigned)tiles, d.C / 64), dim3(256), lds, (hipStream_t)stream, d);
    UD_CHECK_LAUNCH("ud_dwconv7_nhwc_f32 (LDS tile) launch");
    return UD_OK;
  }
  const long long tiles = (long long)d.B * d.H * ((d.W + 7) >> 3);
  dim3 grid((unsigned)((tiles + 3) / 4), (d.C + 255) / 256);
  hipLaunchKernelGGL(dwconv7_kernel, grid, dim3(256), 0, (hipStream_t)stream, d);
  UD_CHECK_LAUNCH("ud_dwconv7_nhwc_f32 launch");
  return UD_OK;
}

extern "C" int ud_layernorm_patchify2(const float* x, void* out, int B, int H, int W, int C, int ldo, float eps, void* stream) {
  if (!x || !out || B <= 0 || H < 2 || W < 2 || C <= 0 || (C & 3) || C > 2048 || ldo < 4 * C || (ldo & 3)) {
    ud_set_error("ud_layernorm_patchify2: bad argument (C % 4 == 0, C <= 2048, ldo >= 4 C)");
    return UD_ERR_BAD_ARG;
  }
  const long long rows = (long long)B * (H >> 1) * 2 * (W >> 1) * 2;
  dim3 grid((unsigned)((rows + 3) / 4));
  hipStream_t s = (hipStream_t)stream;
  if (C <= 256) hipLaunchKernelGGL((ln_patchify2_kernel<1>), grid, dim3(256), 0, s, x, (half_t*)out, B, H, W, C, ldo, eps);
  else if (C <= 512) hipLaunchKernelGGL((ln_patchify2_kernel<2>), grid, dim3(256), 0, s, x, (half_t*)out, B, H, W, C, ldo, eps);
  else if (C <= 1024) hipLaunchKernelGGL((ln_patchify2_kernel<4>), grid, dim3(256), 0, s, x, (half_t*)out, B, H, W, C, ldo, eps);
  else hipLaunchKernelGGL((ln_patchify2_kernel<8>), grid, dim3(256), 0, s, x, (half_t*)out, B, H, W, C, ldo, eps);
  UD_CHECK_LAUNCH("ud_layernorm_patchify2 launch");
  return UD_OK;
}

extern "C" int ud_patchify4_nchw(const float* img, void* out, int B, int H, int W, int ldo, void* stream) {
  if (!img || !out || B <= 0 || H < 4 || W < 4 || ldo < 48) { ud_set_error("ud_patchify4_nchw: bad argument"); return UD_ERR_BAD_ARG; }
  const long long total = (long long)B * (H >> 2) * (W >> 2) * 48;
  long long g = (total + 255) / 256;
  if (g > 256 * 16) g = 256 * 16;
  hipLaunchKernelGGL(patchify4_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, img, (half_t*)out, B, H, W, ldo);
  UD_CHECK_LAUNCH("ud_patchify4_nchw launch");
  return UD_OK;
}

extern "C" int ud_max_f32(float* dst, const float* src, long long n, int init, void* stream) {
  if (!dst || !src || n <= 0 || (n & 3)) { ud_set_error("ud_max_f32: bad argument (n % 4 == 0)"); return UD_ERR_BAD_ARG; }
  long long g = (n / 4 + 255) / 256;
  if (g > 256 * 16) g = 256 * 16;
  hipLaunchKernelGGL(max_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, dst, src, n / 4, init);
  UD_CHECK_LAUNCH("ud_max_f32 launch");
  return UD_OK;
}

extern "C" int ud_spatial_mean_f32(const float* x, float* out, int B, int HW, int C, int ldo, void* stream) {
  if (!x || !out || B <= 0 || HW <= 0 || C <= 0 || (C & 3) || (ldo & 3)) { ud_set_error("ud_spatial_mean_f32: bad argument"); return UD_ERR_BAD_ARG; }
  hipLaunchKernelGGL(spatial_mean_kernel, dim3(B, (C + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, out, HW, C, ldo);
  UD_CHECK_LAUNCH("ud_spatial_mean_f32 launch");
  return UD_OK;
}
