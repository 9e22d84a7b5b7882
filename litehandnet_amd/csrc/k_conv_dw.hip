// Depthwise k x k convolution (stride, dilation) and the 3-channel stem convolution, NHWC fp32.
// HBM-bound: each thread owns 4 channels of one output pixel and gathers its taps through L1/L2
// (neighbouring outputs share them), applying the producer's pending BatchNorm/activation on load;
// the epilogue accumulates this layer's BatchNorm statistics (one double atomic per block+channel).
#include <stdlib.h>
#include "lhn_common.h"

// ------------------------------------------------------------------ depthwise forward
// Block = persistent over output rows (n, ho); thread = (c4, pixel lane).  All index math is 32-bit and the
// row/tap validity is block-uniform; a wave reads 64/C4 neighbouring pixels x C*4 contiguous bytes per tap.
template <int K>
__global__ void __launch_bounds__(256) k_dw_fwd(lhn_view x, const float* __restrict__ w, lhn_view y,
                                                double* __restrict__ stats, int stride, int pad, int dil, lhn_bnfin fin,
                                                int /*unused*/) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int C = x.C, C4 = C >> 2;
  constexpr int KK = K * K;
  float* Ws = smem;                                   // [KK][C]
  f4* red = reinterpret_cast<f4*>(smem + KK * C);     // [256][2]
  const int tid = threadIdx.x;
  const int c4 = tid % C4, pl = tid / C4, PL = 256 / C4;
  const int cin = x.coff + 4 * c4, cout = y.coff + 4 * c4;
  // pending BatchNorm of the input (LDS: >= 8 KB, weights staged after)
  const Xf4 xf = lhn_load_xf(x, cin);
  for (int i = tid; i < KK * C; i += 256) {
    const int c = i / KK, t = i - c * KK;
    Ws[t * C + c] = w ? w[i] : 1.f;
  }
  __syncthreads();
  f4 wt[KK];
#pragma unroll
  for (int t = 0; t < KK; ++t) wt[t] = *reinterpret_cast<const f4*>(Ws + t * C + 4 * c4);
  const int rows = y.N * y.H;
  double sd[4] = {0, 0, 0, 0}, qd[4] = {0, 0, 0, 0};      // per-row fp32 partials promoted to double (see k_conv_pw.hip)
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const int n = row / y.H, ho = row - n * y.H;
    const float* xin = x.data + (size_t)n * x.H * x.W * x.cstride + cin;
    f4 gate = (f4){1.f, 1.f, 1.f, 1.f};
    if (x.gate) gate = *reinterpret_cast<const f4*>(x.gate + (size_t)n * x.cstride + cin);
    float* yout = y.data + (size_t)row * y.W * y.cstride + cout;
    f4 s = (f4){0.f, 0.f, 0.f, 0.f}, q = s, kk = s;     // shifted by the row's first value (see TileStat)
    int cnt = 0;
    for (int wo = LHN_LANE0(pl, PL); wo < y.W; wo += PL) {
      f4 acc = (f4){0.f, 0.f, 0.f, 0.f};
      // branch-free taps: clamped (always in-bounds) addresses, so all K*K loads issue back to back
      f4 raw[KK];
      bool ok[KK];
#pragma unroll
      for (int kh = 0; kh < K; ++kh) {
        const int ih = ho * stride - pad + kh * dil;
        const int ihc = min(max(ih, 0), x.H - 1);
        const float* xr = xin + (size_t)ihc * x.W * x.cstride;
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
          const int iw = wo * stride - pad + kw * dil;
          const int iwc = min(max(iw, 0), x.W - 1);
          ok[kh * K + kw] = (ih == ihc) && (iw == iwc);
          raw[kh * K + kw] = *reinterpret_cast<const f4*>(xr + iwc * x.cstride);
        }
      }
#pragma unroll
      for (int t = 0; t < KK; ++t) {
        const f4 v = lhn_apply_xf(raw[t], xf) * gate;
        acc += (ok[t] ? v : (f4){0.f, 0.f, 0.f, 0.f}) * wt[t];
      }
      *reinterpret_cast<f4*>(yout + wo * y.cstride) = acc;
      if (cnt == 0) kk = acc;
      const f4 d = acc - kk;
      s += d;
      q += d * d;
      ++cnt;
    }
    lhn_unshift4(sd, qd, s, q, kk, cnt);
  }
  if (stats) {
    double* st = stats + (size_t)(blockIdx.x % LHN_STAT_REPLICAS) * 2 * C;
    double* redd = reinterpret_cast<double*>(red);          // [256][2] float4 = 1024 doubles
    if (C4 <= 32 && (C4 & (C4 - 1)) == 0) {
      lhn_block_stat_atomics_d(sd, qd, C4, redd, st, st + C);
    } else {                                                // any other C: one LDS slot per (channel group, pixel lane)
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        redd[tid * 4 + j] = sd[j];
      }
      __syncthreads();
      double ts[4] = {0, 0, 0, 0};
      if (tid < C4)
        for (int j = 0; j < PL; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) ts[e] += redd[(j * C4 + tid) * 4 + e];
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 4; ++j) redd[tid * 4 + j] = qd[j];
      __syncthreads();
      if (tid < C4) {
        double tq[4] = {0, 0, 0, 0};
        for (int j = 0; j < PL; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) tq[e] += redd[(j * C4 + tid) * 4 + e];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          atomicAdd(st + 4 * tid + j, ts[j]);
          atomicAdd(st + C + 4 * tid + j, tq[j]);
        }
      }
    }
    if (fin.counter && lhn_last_block(fin.counter)) lhn_bn_finalize_block(fin, stats);
  }
}

// ------------------------------------------------------------------ stem forward (NCHW 3-channel image -> NHWC)
// thread = (output pixel, group of 8 output channels)
__global__ void __launch_bounds__(256) k_stem_fwd(const float* __restrict__ img, const float* __restrict__ w, lhn_view y,
                                                  double* __restrict__ stats, int Hi, int Wi, int K, int stride, int pad,
                                                  lhn_bnfin fin) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int CO = y.C, CG = CO >> 3, KK = K * K, T = 3 * KK;
  float* Ws = smem;                                  // [T][CO]
  float* red = smem + T * CO;                        // [256][16]
  const int tid = threadIdx.x;
  for (int i = tid; i < T * CO; i += 256) {
    const int co = i / T, t = i - co * T;
    Ws[t * CO + co] = w[i];
  }
  __syncthreads();
  const int cg = tid % CG, pl = tid / CG, PL = 256 / CG;
  const int64_t total = (int64_t)y.N * y.H * y.W;
  float s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
  for (int64_t pix = (int64_t)blockIdx.x * PL + pl; pix < total; pix += (int64_t)gridDim.x * PL) {
    const int wo = (int)(pix % y.W);
    const int64_t t = pix / y.W;
    const int ho = (int)(t % y.H), n = (int)(t / y.H);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int c = 0; c < 3; ++c) {
      const float* plane = img + ((int64_t)n * 3 + c) * Hi * Wi;
      for (int kh = 0; kh < K; ++kh) {
        const int ih = ho * stride - pad + kh;
        if (ih < 0 || ih >= Hi) continue;
        for (int kw = 0; kw < K; ++kw) {
          const int iw = wo * stride - pad + kw;
          if (iw < 0 || iw >= Wi) continue;
          const float v = plane[(int64_t)ih * Wi + iw];
          const float* wr = Ws + (c * KK + kh * K + kw) * CO + cg * 8;
          const f4 w0 = *reinterpret_cast<const f4*>(wr), w1 = *reinterpret_cast<const f4*>(wr + 4);
          acc[0] += v * w0.x; acc[1] += v * w0.y; acc[2] += v * w0.z; acc[3] += v * w0.w;
          acc[4] += v * w1.x; acc[5] += v * w1.y; acc[6] += v * w1.z; acc[7] += v * w1.w;
        }
      }
    }
    float* o = y.data + pix * y.cstride + y.coff + cg * 8;
    *reinterpret_cast<f4*>(o) = (f4){acc[0], acc[1], acc[2], acc[3]};
    *reinterpret_cast<f4*>(o + 4) = (f4){acc[4], acc[5], acc[6], acc[7]};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      s[j] += acc[j];
      q[j] += acc[j] * acc[j];
    }
  }
  if (stats) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[tid * 16 + j] = s[j];
      red[tid * 16 + 8 + j] = q[j];
    }
    __syncthreads();
    if (tid < CO) {
      const int g = tid >> 3, j = tid & 7;
      double sd = 0, qd = 0;
      for (int p = 0; p < PL; ++p) {
        sd += red[(p * CG + g) * 16 + j];
        qd += red[(p * CG + g) * 16 + 8 + j];
      }
      double* st = stats + (size_t)(blockIdx.x % LHN_STAT_REPLICAS) * 2 * CO;
      atomicAdd(st + tid, sd);
      atomicAdd(st + CO + tid, qd);
    }
    if (fin.counter && lhn_last_block(fin.counter)) lhn_bn_finalize_block(fin, stats);
  }
}


// ------------------------------------------------------------------ stem forward, 3x3 -> 32 features (the stems of
// litehourglass.py / liteHandNet.py / lite_hrnet.py) on MFMA: out[pixel][co] = patch[pixel][27 taps] * W^T is a GEMM with
// K = 27 (padded to 32).  The block parks the im2col patch of a 256-pixel tile in LDS (thread = pixel: its 27 image taps are
// loaded ONCE; the direct kernel k_stem_fwd re-reads them per group of 8 features and is bound by 216 LDS weight reads per
// pixel: 125 us for a 184 MB problem); every wave holds all weights as 16 MFMA B registers and runs 2 x 16
// v_mfma_f32_32x32x2_f32 over its 64 pixels.  The accumulator layout gives each lane ONE feature: stores are 128-byte rows,
// statistics need no cross-lane work beyond the two lane halves.  The next tile's taps are loaded while this one computes.
__global__ void __launch_bounds__(256) k_stem3_fwd_mfma(const float* __restrict__ img, const float* __restrict__ w, lhn_view y,
                                                        double* __restrict__ stats, int Hi, int Wi, int stride, int pad, lhn_bnfin fin) {
  constexpr int CO = 32, LDV = 36;
  __shared__ __attribute__((aligned(16))) float Vs[256 * LDV];      // [pixel][tap], columns 27..31 stay zero
  __shared__ double redd[4][2 * CO];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lh = lane >> 5;
  // B fragments: MFMA step ks, lane (n = l31, k = lh) holds W[tap = 16 lh + ks][co = l31]  (the A reads use the same K order)
  float wreg[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    const int tap = 16 * lh + ks;
    wreg[ks] = tap < 27 ? w[l31 * 27 + tap] : 0.f;
  }
  for (int t = 27; t < 32; ++t) Vs[tid * LDV + t] = 0.f;
  const int64_t total = (int64_t)y.N * y.H * y.W;
  const int64_t ntiles = (total + 255) / 256;
  float v[27];
  auto issue = [&](int64_t tile) __attribute__((always_inline)) {
    const int64_t pix = tile * 256 + tid;
    const int64_t pc = pix < total ? pix : total - 1;
    const int wo = (int)(pc % y.W);
    const int64_t r = pc / y.W;
    const int ho = (int)(r % y.H), n = (int)(r / y.H);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* plane = img + ((int64_t)n * 3 + c) * Hi * Wi;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int ih = ho * stride - pad + kh, ihc = min(max(ih, 0), Hi - 1);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int iw = wo * stride - pad + kw, iwc = min(max(iw, 0), Wi - 1);
          const float t = plane[(int64_t)ihc * Wi + iwc];
          v[c * 9 + kh * 3 + kw] = (ih == ihc && iw == iwc) ? t : 0.f;
        }
      }
    }
  };
  float s = 0.f, q = 0.f, kk = 0.f;      // this lane's feature: shifted sums over its pixels (see TileStat)
  int cnt = 0;
  int64_t tile = blockIdx.x;
  if (tile < ntiles) issue(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    __syncthreads();                       // the previous tile's A reads are done
#pragma unroll
    for (int t = 0; t < 27; ++t) Vs[tid * LDV + t] = v[t];
    __syncthreads();
    if (tile + gridDim.x < ntiles) issue(tile + gridDim.x);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int prow = wave * 64 + 32 * h;
      const f4* ap = reinterpret_cast<const f4*>(Vs + (prow + l31) * LDV + 16 * lh);
      const f4 a0 = ap[0], a1 = ap[1], a2 = ap[2], a3 = ap[3];
      const float a[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w};
      f16v acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks], wreg[ks], acc, 0, 0, 0);
      const int64_t pbase = tile * 256 + prow + 4 * lh;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t pix = pbase + (r & 3) + 8 * (r >> 2);
        if (pix < total) {
          y.data[pix * y.cstride + y.coff + l31] = acc[r];
          if (cnt == 0) kk = acc[r];
          const float d = acc[r] - kk;
          s += d;
          q += d * d;
          ++cnt;
        }
      }
    }
  }
  if (stats) {
    const double k0 = kk, c0 = cnt;
    double sd = (double)s + c0 * k0, qd = (double)q + 2.0 * k0 * (double)s + c0 * k0 * k0;
    sd += __shfl_xor(sd, 32, 64);
    qd += __shfl_xor(qd, 32, 64);
    if (lh == 0) {
      redd[wave][l31] = sd;
      redd[wave][CO + l31] = qd;
    }
    __syncthreads();
    if (tid < 2 * CO) {
      double* st = stats + (size_t)(blockIdx.x % LHN_STAT_REPLICAS) * 2 * CO;
      atomicAdd(st + tid, redd[0][tid] + redd[1][tid] + redd[2][tid] + redd[3][tid]);
    }
    if (fin.counter && lhn_last_block(fin.counter)) lhn_bn_finalize_block(fin, stats);
  }
}


// ------------------------------------------------------------------ stem forward, 7x7 -> 64 features (hourglassnet.py:96) on MFMA
// K = 147 taps padded to 160.  64-pixel tiles (42 KB of LDS: 3 blocks per CU); wave = (32-pixel group, 32-feature tile), its
// weights are 80 MFMA B registers.  thread = (pixel, one of 4 tap groups) loads 37 taps; the next tile's are prefetched.
// The direct kernel took 1.9 ms for this layer at batch 64 (147 x 8 scalar FMAs and 2 LDS reads per tap and thread).
__global__ void __launch_bounds__(256, 2) k_stem7_fwd_mfma(const float* __restrict__ img, const float* __restrict__ w, lhn_view y,
                                                        double* __restrict__ stats, int Hi, int Wi, int stride, int pad, lhn_bnfin fin) {
  constexpr int CO = 64, T = 147, TP = 160, NKS = TP / 2, LDV = TP + 4, BMP = 64, G = 4, NR = 6, NV = NR * 7;      // thread = (pixel, tap-row group): rows (c, kh) g, g+4, .. of 21, 7 kw each
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Vs = smem;                                   // [BMP][LDV], columns 147..159 stay zero
  double* redd = reinterpret_cast<double*>(smem + BMP * LDV);      // [2 pixel groups][2 * CO]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lh = lane >> 5;
  const int cot = wave & 1, pg = wave >> 1, co = cot * 32 + l31;
  float wreg[NKS];                                    // step ks, lane (n = l31, k = lh): W[co][tap = NKS * lh + ks]
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int tap = NKS * lh + ks;
    wreg[ks] = tap < T ? w[co * T + tap] : 0.f;
  }
  const int p = tid & (BMP - 1), g = tid >> 6;        // staging role: pixel of the tile, tap group
  for (int t = T + g; t < TP; t += G) Vs[p * LDV + t] = 0.f;
  const int64_t total = (int64_t)y.N * y.H * y.W;
  const int64_t ntiles = (total + BMP - 1) / BMP;
  float v[NV];
  auto issue = [&](int64_t tile) __attribute__((always_inline)) {
    const int64_t pix = tile * BMP + p;
    const int64_t pc = pix < total ? pix : total - 1;
    const int wo = (int)(pc % y.W);
    const int64_t r = pc / y.W;
    const int ho = (int)(r % y.H), n = (int)(r / y.H);
    const float* base = img + (int64_t)n * 3 * Hi * Wi;
    const int iw0 = wo * stride - pad;
#pragma unroll
    for (int jr = 0; jr < NR; ++jr) {
      const int row = min(g + G * jr, 20), c = row / 7, kh = row - 7 * c;
      const int ih = ho * stride - pad + kh, ihc = min(max(ih, 0), Hi - 1);
      const float* rb = base + ((int64_t)c * Hi + ihc) * Wi;
#pragma unroll
      for (int kw = 0; kw < 7; ++kw) {
        const int iw = iw0 + kw, iwc = min(max(iw, 0), Wi - 1);
        const float q = rb[iwc];
        v[jr * 7 + kw] = (ih == ihc && iw == iwc) ? q : 0.f;
      }
    }
  };
  float s = 0.f, q = 0.f, kk = 0.f;                   // this lane's feature: shifted sums (see TileStat)
  int cnt = 0;
  int64_t tile = blockIdx.x;
  if (tile < ntiles) issue(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    __syncthreads();
#pragma unroll
    for (int jr = 0; jr < NR; ++jr)
      if (g + G * jr < 21)
#pragma unroll
        for (int kw = 0; kw < 7; ++kw) Vs[p * LDV + (g + G * jr) * 7 + kw] = v[jr * 7 + kw];
    __syncthreads();
    if (tile + gridDim.x < ntiles) issue(tile + gridDim.x);
    const f4* ap = reinterpret_cast<const f4*>(Vs + (pg * 32 + l31) * LDV + NKS * lh);
    f16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int c = 0; c < NKS / 4; ++c) {
      if ((c & 3) == 0) asm volatile("" ::: "memory");      // at most 4 A fragments (16 registers) in flight: hoisted, all 20 spill
      const f4 a = ap[c];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wreg[4 * c + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wreg[4 * c + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wreg[4 * c + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wreg[4 * c + 3], acc, 0, 0, 0);
    }
    const int64_t pbase = tile * BMP + pg * 32 + 4 * lh;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t pix = pbase + (r & 3) + 8 * (r >> 2);
      if (pix < total) {
        y.data[pix * y.cstride + y.coff + co] = acc[r];
        if (cnt == 0) kk = acc[r];
        const float d = acc[r] - kk;
        s += d;
        q += d * d;
        ++cnt;
      }
    }
  }
  if (stats) {
    const double k0 = kk, c0 = cnt;
    double sd = (double)s + c0 * k0, qd = (double)q + 2.0 * k0 * (double)s + c0 * k0 * k0;
    sd += __shfl_xor(sd, 32, 64);
    qd += __shfl_xor(qd, 32, 64);
    __syncthreads();
    if (lh == 0) {
      redd[pg * 2 * CO + co] = sd;
      redd[pg * 2 * CO + CO + co] = qd;
    }
    __syncthreads();
    if (tid < 2 * CO) {
      double* st = stats + (size_t)(blockIdx.x % LHN_STAT_REPLICAS) * 2 * CO;
      atomicAdd(st + tid, redd[tid] + redd[2 * CO + tid]);
    }
    if (fin.counter && lhn_last_block(fin.counter)) lhn_bn_finalize_block(fin, stats);
  }
}

static inline int grid_for(int64_t items_per_block_total, int per_block, int cap_per_cu) {
  int64_t g = (items_per_block_total + per_block - 1) / per_block;
  const int64_t cap = (int64_t)lhn_num_cus() * cap_per_cu;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}
static inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// One extra input source: the consumed input is coef[0]*value(x) + coef[1]*value(ex.v) (MSRB's `out + ca(cat)`,
// litehourglass.py:41-45, summed while the halo tile is staged instead of being written by an elementwise pass).
struct DwExtra {
  lhn_view v;
  float coef[2];
  int n;             // 0 or 1
  float* sum_out;    // the summed input (tile interiors) is also written here, or NULL (lhn_pw_opts.sum_out)
  int so_cstride, so_coff;
};

int lhn_dwk_fwd_lds(const lhn_view* x, const float* w, const lhn_view* y, double* stats, int k, int dil, lhn_bnfin fin,
                    hipStream_t s, const DwExtra* ex);
struct DwBnSum;
static int dws2_fwd(const lhn_view* x, const float* w, const lhn_view* y, double* stats, lhn_bnfin fin, hipStream_t s);
static int dws2_bwd(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, float* dx, int dx_acc, float* dw,
                    int nrep, int64_t rep_stride, hipStream_t s);
int lhn_dwk_bwd_lds(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, float* dx, int dx_acc,
                    float* dw, int k, int dil, int nrep, int64_t rep_stride, hipStream_t s, const DwBnSum* bs);
static bool lhn_dw_force_gather() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("LHN_DW_GATHER");
    v = (e && e[0] == '1') ? 1 : 0;
  }
  return v == 1;
}

static int dw_fwd_extra(const lhn_view* x, const float* w, const lhn_view* y, double* stats, int k, int stride, int pad, int dil,
                        lhn_bnfin fin, const lhn_view* extra, const float* coef, const lhn_view* so, hipStream_t s);

extern "C" int lhn_conv_dw_fwd3(const lhn_view* x, const float* w, const lhn_view* y, double* stats, int k, int stride,
                                int pad, int dil, const lhn_bnfin* finp, const lhn_view* extra, const float* coef2,
                                const lhn_view* sum_out, void* stream) {
  if (!extra) {
    LHN_CHECK_ARG(!sum_out, "lhn_conv_dw_fwd3: sum_out without a second source");
    return lhn_conv_dw_fwd(x, w, y, stats, k, stride, pad, dil, finp, stream);
  }
  lhn_bnfin fin;
  if (finp && stats) fin = *finp; else fin.counter = nullptr;
  LHN_CHECK_ARG(lhn_view_ok(x) && lhn_view_ok(y) && lhn_view_ok(extra) && w && coef2, "lhn_conv_dw_fwd2: bad view / null pointer");
  LHN_CHECK_ARG(extra->C == x->C && extra->N == x->N && extra->H == x->H && extra->W == x->W, "lhn_conv_dw_fwd2: extra source geometry");
  LHN_CHECK_ARG(x->C == y->C && y->N == x->N && y->H == x->H && y->W == x->W, "lhn_conv_dw_fwd2: same-size output");
  LHN_CHECK_ARG(!sum_out || (sum_out->data && sum_out->C == x->C && sum_out->N == x->N && sum_out->H == x->H && sum_out->W == x->W &&
                             sum_out->cstride % 4 == 0 && sum_out->coff % 4 == 0 && sum_out->coff + sum_out->C <= sum_out->cstride),
                "lhn_conv_dw_fwd3: sum_out geometry (same pixels and channels as x)");
  const int rc = dw_fwd_extra(x, w, y, stats, k, stride, pad, dil, fin, extra, coef2, sum_out, (hipStream_t)stream);
  LHN_CHECK_ARG(rc == 1, "lhn_conv_dw_fwd2: a second source needs k=3, stride 1, 'same' padding, W >= 8 (got k=%d s=%d C=%d W=%d)",
                k, stride, x->C, y->W);
  LHN_CHECK_LAUNCH("lhn_conv_dw_fwd2");
  return 0;
}

extern "C" int lhn_conv_dw_fwd2(const lhn_view* x, const float* w, const lhn_view* y, double* stats, int k, int stride,
                                int pad, int dil, const lhn_bnfin* finp, const lhn_view* extra, const float* coef2, void* stream) {
  return lhn_conv_dw_fwd3(x, w, y, stats, k, stride, pad, dil, finp, extra, coef2, nullptr, stream);
}

extern "C" int lhn_conv_dw_fwd(const lhn_view* x, const float* w, const lhn_view* y, double* stats, int k, int stride,
                               int pad, int dil, const lhn_bnfin* finp, void* stream) {
  lhn_bnfin fin;
  if (finp && stats) fin = *finp; else fin.counter = nullptr;
  LHN_CHECK_ARG(lhn_view_ok(x) && lhn_view_ok(y) && (w || k == 1), "lhn_conv_dw_fwd: bad view / null pointer (w may be NULL = ones only for k=1)");
  LHN_CHECK_ARG(x->C == y->C && x->C % 4 == 0 && x->C <= 512, "lhn_conv_dw_fwd: channels %d -> %d (multiple of 4, <= 512)", x->C, y->C);
  LHN_CHECK_ARG((k == 1 || k == 3 || k == 5 || k == 7) && stride >= 1 && dil >= 1 && pad >= 0, "lhn_conv_dw_fwd: k=%d stride=%d dil=%d", k, stride, dil);
  const int Ho = (x->H + 2 * pad - dil * (k - 1) - 1) / stride + 1, Wo = (x->W + 2 * pad - dil * (k - 1) - 1) / stride + 1;
  LHN_CHECK_ARG(y->N == x->N && y->H == Ho && y->W == Wo, "lhn_conv_dw_fwd: output %dx%d, expected %dx%d", y->H, y->W, Ho, Wo);
  const size_t lds = (size_t)(k * k * x->C) * 4 + 256 * 2 * 16;
  const int grid = grid_for((int64_t)y->N * Ho, 1, 8);
  hipStream_t s = (hipStream_t)stream;
  LHN_CHECK_ARG(lhn_no_pend(x), "lhn_conv_dw_fwd: lhn_view.pend is reserved (NULL)");
  const int px = 0;
  DwExtra ex0;
  ex0.n = 0;
  ex0.sum_out = nullptr;
  if (w && stride == 1 && pad == dil * (k - 1) / 2 && x->C % 4 == 0 && y->W >= 8 && !lhn_dw_force_gather() &&
      lhn_dwk_fwd_lds(x, w, y, stats, k, dil, fin, s, &ex0)) {
  } else if (w && k == 3 && stride == 2 && pad == 1 && dil == 1 && x->C % 4 == 0 && !lhn_dw_force_gather() &&
             dws2_fwd(x, w, y, stats, fin, s)) {
  } else if (k == 3)
    hipLaunchKernelGGL((k_dw_fwd<3>), dim3(grid), dim3(256), lds, s, *x, w, *y, stats, stride, pad, dil, fin, px);
  else if (k == 7)
    hipLaunchKernelGGL((k_dw_fwd<7>), dim3(grid), dim3(256), lds, s, *x, w, *y, stats, stride, pad, dil, fin, px);
  else if (k == 1)
    hipLaunchKernelGGL((k_dw_fwd<1>), dim3(grid), dim3(256), lds, s, *x, w, *y, stats, stride, pad, dil, fin, px);
  else if (k == 5)
    hipLaunchKernelGGL((k_dw_fwd<5>), dim3(grid), dim3(256), lds, s, *x, w, *y, stats, stride, pad, dil, fin, px);
  LHN_CHECK_LAUNCH("lhn_conv_dw_fwd");
  return 0;
}

extern "C" int lhn_conv_stem_fwd(const float* img, const float* w, const lhn_view* y, double* stats, int Hi, int Wi, int k,
                                 int stride, int pad, const lhn_bnfin* finp, void* stream) {
  lhn_bnfin fin;
  if (finp && stats) fin = *finp; else fin.counter = nullptr;
  LHN_CHECK_ARG(img && w && lhn_view_ok(y), "lhn_conv_stem_fwd: bad view / null pointer");
  LHN_CHECK_ARG(y->C % 8 == 0 && pow2(y->C / 8) && y->C <= 256, "lhn_conv_stem_fwd: Cout=%d", y->C);
  LHN_CHECK_ARG((k == 1 || k == 3 || k == 5 || k == 7) && stride >= 1, "lhn_conv_stem_fwd: k=%d", k);
  const int Ho = (Hi + 2 * pad - k) / stride + 1, Wo = (Wi + 2 * pad - k) / stride + 1;
  LHN_CHECK_ARG(y->H == Ho && y->W == Wo, "lhn_conv_stem_fwd: output %dx%d, expected %dx%d", y->H, y->W, Ho, Wo);
  const int PL = 256 / (y->C / 8);
  const size_t lds = (size_t)(3 * k * k * y->C) * 4 + 256 * 16 * 4;
  if (k == 7 && y->C == 64 && !lhn_dw_force_gather()) {
    const size_t lds7 = (size_t)64 * 164 * 4 + 2 * 2 * 64 * 8;
    static LhnKernelCfg cfg7;
    int per_cu = 1;
    if (!lhn_kernel_cfg(cfg7, &k_stem7_fwd_mfma, lds7, 3, &per_cu)) {
      lhn_set_error("lhn_conv_stem_fwd: cannot reserve %zu B of LDS", lds7);
      return 2;
    }
    const int64_t ntiles = ((int64_t)y->N * Ho * Wo + 63) / 64;
    int grid = lhn_num_cus() * per_cu;
    if (grid > ntiles) grid = (int)ntiles;
    hipLaunchKernelGGL(k_stem7_fwd_mfma, dim3(grid), dim3(256), lds7, (hipStream_t)stream, img, w, *y, stats, Hi, Wi, stride, pad, fin);
  } else if (k == 3 && y->C == 32 && !lhn_dw_force_gather()) {
    // (3 / 6 / 12 blocks per CU measured in round 3: no difference, forward 2.680 / 2.677 / 2.682 ms)
    hipLaunchKernelGGL(k_stem3_fwd_mfma, dim3(grid_for((int64_t)y->N * Ho * Wo, 256, 3)), dim3(256), 0, (hipStream_t)stream, img, w, *y,
                       stats, Hi, Wi, stride, pad, fin);
  } else
    hipLaunchKernelGGL(k_stem_fwd, dim3(grid_for((int64_t)y->N * Ho * Wo, PL, 8)), dim3(256), lds, (hipStream_t)stream, img,
                       w, *y, stats, Hi, Wi, k, stride, pad, fin);
  LHN_CHECK_LAUNCH("lhn_conv_stem_fwd");
  return 0;
}

// =====================================================================================================
// Backward.  dy is formed on the fly:  du = lrelu'(u) * (gate*dz + pooled-gradient),  dy = A*du + B*y + C.
__device__ __forceinline__ f4 dw_load_dy(const lhn_view& y, const lhn_gradview& g, const Xf4& xf, const Gr4& gr,
                                         int64_t pix, int n, int h, int w, int ca) {
  const f4 raw = *reinterpret_cast<const f4*>(y.data + pix * y.cstride + ca);
  const f4 dz = *reinterpret_cast<const f4*>(g.dz + pix * y.cstride + ca);
  const f4 du = lhn_grad_du(y, g, xf, raw, dz, n, h, w, ca);
  return gr.A * du + gr.B * raw + gr.Cc;
}

// row-local dy loader: off = element offset of (n, h, w, channel ca) inside the y buffer
__device__ __forceinline__ f4 dw_dy_at(const lhn_view& y, const lhn_gradview& g, const Xf4& xf, const Gr4& gr, f4 gate,
                                       size_t off, int n, int h, int w, int ca) {
  const f4 raw = *reinterpret_cast<const f4*>(y.data + off);
  f4 e = *reinterpret_cast<const f4*>(g.dz + off) * gate;
  const f4 u = raw * xf.sc + xf.sh;
  const f4 dl = (f4){u.x > 0.f ? 1.f : xf.sl.x, u.y > 0.f ? 1.f : xf.sl.y, u.z > 0.f ? 1.f : xf.sl.z, u.w > 0.f ? 1.f : xf.sl.w};
  if (g.dpool) e += lhn_dpool_sum(g, y, n, h, w, ca);
  const f4 du = e * dl;
  return gr.A * du + gr.B * raw + gr.Cc;
}

// dgrad: block = persistent over INPUT rows (n, hi); thread = (c4, pixel lane)
template <int K>
__global__ void __launch_bounds__(256) k_dw_bwd_data(lhn_view x, const float* __restrict__ w, lhn_view y, lhn_gradview gy,
                                                     float* __restrict__ dx, int accumulate, int stride, int pad, int dil) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int C = x.C, C4 = C >> 2;
  constexpr int KK = K * K;
  float* Ws = smem;
  const int tid = threadIdx.x;
  for (int i = tid; i < KK * C; i += 256) {
    const int c = i / KK, t = i - c * KK;
    Ws[t * C + c] = w ? w[i] : 1.f;
  }
  __syncthreads();
  const int c4 = tid % C4, pl = tid / C4, PL = 256 / C4;
  const int cx = x.coff + 4 * c4, cy = y.coff + 4 * c4;
  const Xf4 yxf = lhn_load_xf(y, cy);
  const Gr4 ygr = lhn_load_coef(gy, y.cstride, cy);
  f4 wt[KK];
#pragma unroll
  for (int t = 0; t < KK; ++t) wt[t] = *reinterpret_cast<const f4*>(Ws + t * C + 4 * c4);
  const int rows = x.N * x.H;
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const int n = row / x.H, hi = row - n * x.H;
    f4 gate = (f4){1.f, 1.f, 1.f, 1.f};
    if (y.gate) gate = *reinterpret_cast<const f4*>(y.gate + (size_t)n * y.cstride + cy);
    float* dxr = dx + (size_t)row * x.W * x.cstride + cx;
    for (int wi = LHN_LANE0(pl, PL); wi < x.W; wi += PL) {
      f4 acc = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kh = 0; kh < K; ++kh) {
        const int hn = hi + pad - kh * dil;
        if (hn < 0 || (stride > 1 && hn % stride)) continue;
        const int ho = hn / stride;
        if (ho >= y.H) continue;
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
          const int wn = wi + pad - kw * dil;
          if (wn < 0 || (stride > 1 && wn % stride)) continue;
          const int wo = wn / stride;
          if (wo >= y.W) continue;
          const size_t off = ((size_t)(n * y.H + ho) * y.W + wo) * y.cstride + cy;
          acc += dw_dy_at(y, gy, yxf, ygr, gate, off, n, ho, wo, cy) * wt[kh * K + kw];
        }
      }
      float* o = dxr + wi * x.cstride;
      if (accumulate) acc += *reinterpret_cast<const f4*>(o);
      *reinterpret_cast<f4*>(o) = acc;
    }
  }
}

// wgrad: block = persistent over OUTPUT rows; thread = (c4, pixel lane); KR kernel rows [kh0, kh0+KR) per launch
template <int K, int KR>
__global__ void __launch_bounds__(256) k_dw_bwd_weight(lhn_view x, lhn_view y, lhn_gradview gy, float* __restrict__ dw,
                                                       int stride, int pad, int dil, int kh0, int nrep, int64_t rep_stride) {
  dw += (size_t)(blockIdx.x % nrep) * rep_stride;
  __shared__ f4 red[256];
  const int C4 = x.C >> 2;
  const int tid = threadIdx.x, c4 = tid % C4, pl = tid / C4, PL = 256 / C4;
  const int cx = x.coff + 4 * c4, cy = y.coff + 4 * c4;
  const Xf4 xxf = lhn_load_xf(x, cx), yxf = lhn_load_xf(y, cy);
  const Gr4 ygr = lhn_load_coef(gy, y.cstride, cy);
  f4 acc[KR * K];
#pragma unroll
  for (int i = 0; i < KR * K; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
  const int rows = y.N * y.H;
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const int n = row / y.H, ho = row - n * y.H;
    f4 ygate = (f4){1.f, 1.f, 1.f, 1.f}, xgate = ygate;
    if (y.gate) ygate = *reinterpret_cast<const f4*>(y.gate + (size_t)n * y.cstride + cy);
    if (x.gate) xgate = *reinterpret_cast<const f4*>(x.gate + (size_t)n * x.cstride + cx);
    const float* xin = x.data + (size_t)n * x.H * x.W * x.cstride + cx;
    for (int wo = LHN_LANE0(pl, PL); wo < y.W; wo += PL) {
      const size_t off = ((size_t)row * y.W + wo) * y.cstride + cy;
      f4 raw[KR * K];
      bool ok[KR * K];
#pragma unroll
      for (int r = 0; r < KR; ++r) {
        const int ih = ho * stride - pad + (kh0 + r) * dil;
        const int ihc = min(max(ih, 0), x.H - 1);
        const float* xr = xin + (size_t)ihc * x.W * x.cstride;
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
          const int iw = wo * stride - pad + kw * dil;
          const int iwc = min(max(iw, 0), x.W - 1);
          ok[r * K + kw] = (ih == ihc) && (iw == iwc);
          raw[r * K + kw] = *reinterpret_cast<const f4*>(xr + iwc * x.cstride);
        }
      }
      const f4 dy = dw_dy_at(y, gy, yxf, ygr, ygate, off, n, ho, wo, cy);
#pragma unroll
      for (int t = 0; t < KR * K; ++t) {
        const f4 v = lhn_apply_xf(raw[t], xxf) * xgate;
        acc[t] += dy * (ok[t] ? v : (f4){0.f, 0.f, 0.f, 0.f});
      }
    }
  }
#pragma unroll
  for (int i = 0; i < KR * K; ++i) {
    __syncthreads();
    red[tid] = acc[i];
    __syncthreads();
    if (tid < C4) {
      f4 s = (f4){0.f, 0.f, 0.f, 0.f};
      for (int j = 0; j < PL; ++j) s += red[j * C4 + tid];
      const int tap = (kh0 + i / K) * K + (i % K);
      atomicAdd(dw + (4 * tid + 0) * K * K + tap, s.x);
      atomicAdd(dw + (4 * tid + 1) * K * K + tap, s.y);
      atomicAdd(dw + (4 * tid + 2) * K * K + tap, s.z);
      atomicAdd(dw + (4 * tid + 3) * K * K + tap, s.w);
    }
  }
}

// stem wgrad: thread = (pixel lane, 4 output channels); T = 3*KR*K accumulators of float4 for the kernel rows
// [kh0, kh0 + KR) (K = 7, hourglassnet.py:100, runs one kernel row per launch: 147 float4 accumulators do not fit)
template <int K, int KR>
__global__ void __launch_bounds__(256) k_stem_bwd(const float* __restrict__ img, lhn_view y, lhn_gradview gy,
                                                  float* __restrict__ dw, int Hi, int Wi, int stride, int pad, int nrep,
                                                  int64_t rep_stride, int kh0) {
  dw += (size_t)(blockIdx.x % nrep) * rep_stride;
  constexpr int T = 3 * KR * K;
  __shared__ f4 red[256];
  const int C4 = y.C >> 2;
  const int tid = threadIdx.x, c4 = tid % C4, pl = tid / C4, PL = 256 / C4;
  const int cy = y.coff + 4 * c4;
  const Xf4 yxf = lhn_load_xf(y, cy);
  const Gr4 ygr = lhn_load_coef(gy, y.cstride, cy);
  f4 acc[T];
#pragma unroll
  for (int i = 0; i < T; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
  const int64_t total = (int64_t)y.N * y.H * y.W;
  for (int64_t pix = (int64_t)blockIdx.x * PL + pl; pix < total; pix += (int64_t)gridDim.x * PL) {
    const int wo = (int)(pix % y.W);
    const int64_t t = pix / y.W;
    const int ho = (int)(t % y.H), n = (int)(t / y.H);
    const f4 dy = dw_load_dy(y, gy, yxf, ygr, pix, n, ho, wo, cy);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* plane = img + ((int64_t)n * 3 + c) * Hi * Wi;
#pragma unroll
      for (int r = 0; r < KR; ++r) {
        const int ih = ho * stride - pad + kh0 + r;
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
          const int iw = wo * stride - pad + kw;
          float v = 0.f;
          if (ih >= 0 && ih < Hi && iw >= 0 && iw < Wi) v = plane[(int64_t)ih * Wi + iw];
          acc[(c * KR + r) * K + kw] += dy * v;
        }
      }
    }
  }
  constexpr int TW = 3 * K * K;            // weights per output channel
#pragma unroll
  for (int i = 0; i < T; ++i) {
    __syncthreads();
    red[tid] = acc[i];
    __syncthreads();
    if (tid < C4) {
      f4 s = (f4){0.f, 0.f, 0.f, 0.f};
      for (int j = 0; j < PL; ++j) s += red[j * C4 + tid];
      const int c = i / (KR * K), r = (i / K) % KR, kw = i % K, e = c * K * K + (kh0 + r) * K + kw;
      atomicAdd(dw + (4 * tid + 0) * TW + e, s.x);
      atomicAdd(dw + (4 * tid + 1) * TW + e, s.y);
      atomicAdd(dw + (4 * tid + 2) * TW + e, s.z);
      atomicAdd(dw + (4 * tid + 3) * TW + e, s.w);
    }
  }
}


// ------------------------------------------------------------------ stem weight gradient, 3x3 -> 32 features, on MFMA:
// dW[co][tap] = sum_pixels dy[pixel][co] * patch[pixel][tap] is a (32 x 27) = dY^T (32 x P) * V (P x 27) GEMM with the
// pixels as K.  Per 256-pixel tile the block parks dy (formed on the fly from dz, raw y and the BatchNorm-backward
// coefficients) and the im2col patch (27 taps, padded to 32 zero columns) in LDS; every wave then runs 32
// v_mfma_f32_32x32x2_f32 over its 64 pixels into one 32x32 accumulator that lives across the whole launch.
__global__ void __launch_bounds__(256) k_stem3_bwd_mfma(const float* __restrict__ img, lhn_view y, lhn_gradview gy, float* __restrict__ dw,
                                                        int Hi, int Wi, int stride, int pad, int nrep, int64_t rep_stride) {
  constexpr int CO = 32, LDY = CO + 4, LDV = 33;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* dYs = smem;                     // [256][LDY] pixel-major dy; reused for the final cross-wave sum (4096 floats)
  float* Vs = smem + 256 * LDY;          // [256][LDV] pixel-major patch, columns 27..31 stay zero
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lh = lane >> 5;
  const int c4 = tid & 7, prow = tid >> 3;
  const int cy = y.coff + 4 * c4;
  const Xf4 yxf = lhn_load_xf(y, cy);
  const Gr4 ygr = lhn_load_coef(gy, y.cstride, cy);
  for (int t = 27; t < 32; ++t) Vs[tid * LDV + t] = 0.f;
  f16v acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int64_t total = (int64_t)y.N * y.H * y.W;
  const int64_t ntiles = (total + 255) / 256;
  // raw loads of a tile go into registers one tile AHEAD (dy is formed when they are parked in LDS)
  f4 yraw[8], zraw[8];
  float v[27];
  auto issue = [&](int64_t tile) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int64_t pix = tile * 256 + prow + 32 * j;
      pix = pix < total ? pix : total - 1;
      yraw[j] = *reinterpret_cast<const f4*>(y.data + pix * y.cstride + cy);
      zraw[j] = *reinterpret_cast<const f4*>(gy.dz + pix * y.cstride + cy);
    }
    const int64_t pix = tile * 256 + tid;
    const int64_t pc = pix < total ? pix : total - 1;
    const int wo = (int)(pc % y.W);
    const int64_t r = pc / y.W;
    const int ho = (int)(r % y.H), n = (int)(r / y.H);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* plane = img + ((int64_t)n * 3 + c) * Hi * Wi;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int ih = ho * stride - pad + kh, ihc = min(max(ih, 0), Hi - 1);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int iw = wo * stride - pad + kw, iwc = min(max(iw, 0), Wi - 1);
          const float t = plane[(int64_t)ihc * Wi + iwc];
          v[c * 9 + kh * 3 + kw] = (ih == ihc && iw == iwc) ? t : 0.f;      // (rows past the end meet dy = 0)
        }
      }
    }
  };
  int64_t tile = blockIdx.x;
  if (tile < ntiles) issue(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    __syncthreads();                       // the previous tile's MFMA reads are done
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int p = prow + 32 * j;
      const int64_t pix = tile * 256 + p;
      f4 d = (f4){0.f, 0.f, 0.f, 0.f};
      if (pix < total) {
        const int wo = (int)(pix % y.W);
        const int64_t r = pix / y.W;
        const f4 du = lhn_grad_du(y, gy, yxf, yraw[j], zraw[j], (int)(r / y.H), (int)(r % y.H), wo, cy);
        d = ygr.A * du + ygr.B * yraw[j] + ygr.Cc;
      }
      *reinterpret_cast<f4*>(dYs + p * LDY + 4 * c4) = d;
    }
#pragma unroll
    for (int t = 0; t < 27; ++t) Vs[tid * LDV + t] = v[t];
    __syncthreads();
    if (tile + gridDim.x < ntiles) issue(tile + gridDim.x);
    const float* ar = dYs + (wave * 64 + lh) * LDY + l31;
    const float* br = Vs + (wave * 64 + lh) * LDV + l31;
#pragma unroll 8
    for (int ks = 0; ks < 32; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[2 * ks * LDY], br[2 * ks * LDV], acc, 0, 0, 0);
  }
  // cross-wave sum through LDS, then one float atomic per weight into this block's gradient replica
  __syncthreads();
  float* part = dYs;                                 // [4][16][64]
#pragma unroll
  for (int r = 0; r < 16; ++r) part[(wave * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
  dw += (size_t)(blockIdx.x % nrep) * rep_stride;
  for (int i = tid; i < CO * 27; i += 256) {
    const int co = i / 27, t = i - co * 27;
    const int r = (co & 3) + 4 * (co >> 3), ln = t + 32 * ((co >> 2) & 1);      // accumulator row co = (r & 3) + 8 (r >> 2) + 4 lh
    const float v = part[(0 * 16 + r) * 64 + ln] + part[(1 * 16 + r) * 64 + ln] + part[(2 * 16 + r) * 64 + ln] + part[(3 * 16 + r) * 64 + ln];
    atomicAdd(dw + i, v);
  }
}


// ------------------------------------------------------------------ stem weight gradient, 7x7 -> 64 features, on MFMA:
// dW (64 x 147) = dY^T (64 x P) * patch (P x 147): 2 feature tiles x 5 tap tiles of 32 x 32 over the four waves (3, 3, 2, 2),
// K = the 64 pixels of a tile.  Staging as in k_stem7_fwd_mfma (row-wise taps) + k_stem3_bwd_mfma (dy formed at commit).
__global__ void __launch_bounds__(256, 2) k_stem7_bwd_mfma(const float* __restrict__ img, lhn_view y, lhn_gradview gy, float* __restrict__ dw,
                                                           int Hi, int Wi, int stride, int pad, int nrep, int64_t rep_stride) {
  constexpr int CO = 64, T = 147, LDY = CO + 4, LDV = 161, BMP = 64, G = 4, NR = 6;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* dYs = smem;                     // [BMP][LDY]
  float* Vs = smem + BMP * LDY;          // [BMP][LDV], columns 147..159 stay zero
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lh = lane >> 5;
  const int c4 = tid & 15, prow = tid >> 4;
  const int cy = y.coff + 4 * c4;
  const Xf4 yxf = lhn_load_xf(y, cy);
  const Gr4 ygr = lhn_load_coef(gy, y.cstride, cy);
  const int p = tid & (BMP - 1), g = tid >> 6;
  for (int t = T + g; t < 160; t += G) Vs[p * LDV + t] = 0.f;
  f16v acc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int64_t total = (int64_t)y.N * y.H * y.W;
  const int64_t ntiles = (total + BMP - 1) / BMP;
  f4 yraw[4], zraw[4];
  float v[NR * 7];
  auto issue = [&](int64_t tile) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int64_t pix = tile * BMP + prow + 16 * j;
      pix = pix < total ? pix : total - 1;
      yraw[j] = *reinterpret_cast<const f4*>(y.data + pix * y.cstride + cy);
      zraw[j] = *reinterpret_cast<const f4*>(gy.dz + pix * y.cstride + cy);
    }
    const int64_t pix = tile * BMP + p;
    const int64_t pc = pix < total ? pix : total - 1;
    const int wo = (int)(pc % y.W);
    const int64_t r = pc / y.W;
    const int ho = (int)(r % y.H), n = (int)(r / y.H);
    const float* base = img + (int64_t)n * 3 * Hi * Wi;
    const int iw0 = wo * stride - pad;
#pragma unroll
    for (int jr = 0; jr < NR; ++jr) {
      const int row = min(g + G * jr, 20), c = row / 7, kh = row - 7 * c;
      const int ih = ho * stride - pad + kh, ihc = min(max(ih, 0), Hi - 1);
      const float* rb = base + ((int64_t)c * Hi + ihc) * Wi;
#pragma unroll
      for (int kw = 0; kw < 7; ++kw) {
        const int iw = iw0 + kw, iwc = min(max(iw, 0), Wi - 1);
        const float q = rb[iwc];
        v[jr * 7 + kw] = (ih == ihc && iw == iwc) ? q : 0.f;
      }
    }
  };
  int64_t tile = blockIdx.x;
  if (tile < ntiles) issue(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pp = prow + 16 * j;
      const int64_t pix = tile * BMP + pp;
      f4 d = (f4){0.f, 0.f, 0.f, 0.f};
      if (pix < total) {
        const int wo = (int)(pix % y.W);
        const int64_t r = pix / y.W;
        const f4 du = lhn_grad_du(y, gy, yxf, yraw[j], zraw[j], (int)(r / y.H), (int)(r % y.H), wo, cy);
        d = ygr.A * du + ygr.B * yraw[j] + ygr.Cc;
      }
      *reinterpret_cast<f4*>(dYs + pp * LDY + 4 * c4) = d;
    }
#pragma unroll
    for (int jr = 0; jr < NR; ++jr)
      if (g + G * jr < 21)
#pragma unroll
        for (int kw = 0; kw < 7; ++kw) Vs[p * LDV + (g + G * jr) * 7 + kw] = v[jr * 7 + kw];
    __syncthreads();
    if (tile + gridDim.x < ntiles) issue(tile + gridDim.x);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int job = wave + 4 * i;                  // 10 jobs: feature tile = job & 1, tap tile = job >> 1
      if (job < 10) {
        const float* ar = dYs + lh * LDY + (job & 1) * 32 + l31;
        const float* br = Vs + lh * LDV + (job >> 1) * 32 + l31;
#pragma unroll 8
        for (int ks = 0; ks < 32; ++ks) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[2 * ks * LDY], br[2 * ks * LDV], acc[i], 0, 0, 0);
      }
    }
  }
  dw += (size_t)(blockIdx.x % nrep) * rep_stride;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int job = wave + 4 * i;
    if (job < 10) {
      const int t = (job >> 1) * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = (job & 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (t < T) atomicAdd(dw + co * T + t, acc[i][r]);
      }
    }
  }
}

static int dw_bwd_fused(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, float* dx, float* dw, int k,
                        int dil, int nrep, int64_t rep_stride, double* bn_sums, const float* bn_save, int bn_C, int bn_coff, hipStream_t s);

extern "C" int lhn_conv_dw_bwd2(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, float* dx,
                                int dx_accumulate, float* dw, int k, int stride, int pad, int dil, int nrep, int64_t rep_stride,
                                double* bn_sums, const float* bn_save, int bn_C, int bn_coff, void* stream) {
  if (!bn_sums) return lhn_conv_dw_bwd(x, w, y, gy, dx, dx_accumulate, dw, k, stride, pad, dil, nrep, rep_stride, stream);
  if (nrep < 1) nrep = 1;
  LHN_CHECK_ARG(lhn_view_ok(x) && lhn_view_ok(y) && gy && gy->dz && w && dw && dx && bn_save && lhn_no_pend(x) && lhn_no_pend(y),
                "lhn_conv_dw_bwd2: bad view / null pointer");
  LHN_CHECK_ARG(!dx_accumulate && !x->gate && stride == 1 && pad == dil * (k - 1) / 2 && x->C % 32 == 0 && x->W >= 8 && x->C == y->C &&
                    bn_coff >= 0 && bn_coff + x->C <= bn_C,
                "lhn_conv_dw_bwd2: fused BatchNorm sums need this convolution to be the only, ungated, stride-1 reader of x (dx stored)");
  const int rc = dw_bwd_fused(x, w, y, gy, dx, dw, k, dil, nrep, rep_stride, bn_sums, bn_save, bn_C, bn_coff, (hipStream_t)stream);
  LHN_CHECK_ARG(rc == 1, "lhn_conv_dw_bwd2: fused BatchNorm sums are built for the 3x3 / dilation 1 kernel (k=%d dil=%d)", k, dil);
  LHN_CHECK_LAUNCH("lhn_conv_dw_bwd2");
  return 0;
}

static int dw_bwd_addends(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, float* dx, int dx_acc, float* dw,
                          int k, int dil, int nrep, int64_t rep_stride, const float* a0, const float* a1, hipStream_t s);

extern "C" int lhn_conv_dw_bwd3(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, float* dx,
                                int dx_accumulate, float* dw, int k, int stride, int pad, int dil, int nrep, int64_t rep_stride,
                                const float* dx_add0, const float* dx_add1, void* stream) {
  if (!dx_add0 && !dx_add1) return lhn_conv_dw_bwd(x, w, y, gy, dx, dx_accumulate, dw, k, stride, pad, dil, nrep, rep_stride, stream);
  if (nrep < 1) nrep = 1;
  LHN_CHECK_ARG(lhn_view_ok(x) && lhn_view_ok(y) && gy && gy->dz && w && dw && dx && lhn_no_pend(x) && lhn_no_pend(y),
                "lhn_conv_dw_bwd3: bad view / null pointer");
  LHN_CHECK_ARG(stride == 1 && pad == dil * (k - 1) / 2 && x->C % 4 == 0 && x->W >= 8 && x->C == y->C,
                "lhn_conv_dw_bwd3: gradient addends need the tiled stride-1 kernel (W >= 8)");
  const int rc = dw_bwd_addends(x, w, y, gy, dx, dx_accumulate, dw, k, dil, nrep, rep_stride, dx_add0 ? dx_add0 : dx_add1,
                                dx_add0 ? dx_add1 : nullptr, (hipStream_t)stream);
  LHN_CHECK_ARG(rc == 1, "lhn_conv_dw_bwd3: no tiled kernel for k=%d dil=%d", k, dil);
  LHN_CHECK_LAUNCH("lhn_conv_dw_bwd3");
  return 0;
}

extern "C" int lhn_conv_dw_bwd(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, float* dx,
                               int dx_accumulate, float* dw, int k, int stride, int pad, int dil, int nrep, int64_t rep_stride,
                               void* stream) {
  if (nrep < 1) nrep = 1;
  LHN_CHECK_ARG(lhn_view_ok(x) && lhn_view_ok(y) && gy && gy->dz && ((w && dw) || k == 1) && lhn_no_pend(x) && lhn_no_pend(y), "lhn_conv_dw_bwd: bad view / null pointer");
  LHN_CHECK_ARG(x->C == y->C && x->C % 4 == 0 && x->C <= 512, "lhn_conv_dw_bwd: channels");
  LHN_CHECK_ARG(k == 1 || k == 3 || k == 7, "lhn_conv_dw_bwd: k=%d (1, 3 or 7)", k);
  hipStream_t s = (hipStream_t)stream;
  if (w && dw && stride == 1 && pad == dil * (k - 1) / 2 && x->C % 4 == 0 && x->W >= 8 && !lhn_dw_force_gather() &&
      lhn_dwk_bwd_lds(x, w, y, gy, dx, dx_accumulate, dw, k, dil, nrep, rep_stride, s, nullptr)) {
    LHN_CHECK_LAUNCH("lhn_conv_dw_bwd");
    return 0;
  }
  if (w && dw && k == 3 && stride == 2 && pad == 1 && dil == 1 && x->C % 4 == 0 && !lhn_dw_force_gather() &&
      dws2_bwd(x, w, y, gy, dx, dx_accumulate, dw, nrep, rep_stride, s)) {
    LHN_CHECK_LAUNCH("lhn_conv_dw_bwd");
    return 0;
  }
  if (dx) {
    const size_t lds = (size_t)(k * k * x->C) * 4;
    const int g = grid_for((int64_t)x->N * x->H, 1, 8);
    if (k == 3)
      hipLaunchKernelGGL((k_dw_bwd_data<3>), dim3(g), dim3(256), lds, s, *x, w, *y, *gy, dx, dx_accumulate, stride, pad, dil);
    else if (k == 1)
      hipLaunchKernelGGL((k_dw_bwd_data<1>), dim3(g), dim3(256), lds, s, *x, w, *y, *gy, dx, dx_accumulate, stride, pad, dil);
    else
      hipLaunchKernelGGL((k_dw_bwd_data<7>), dim3(g), dim3(256), lds, s, *x, w, *y, *gy, dx, dx_accumulate, stride, pad, dil);
  }
  const int gw = grid_for((int64_t)y->N * y->H, 1, 4);
  if (!dw) {
    // identity depthwise (no weight to learn)
  } else if (k == 1) {
    hipLaunchKernelGGL((k_dw_bwd_weight<1, 1>), dim3(gw), dim3(256), 0, s, *x, *y, *gy, dw, stride, pad, dil, 0, nrep, rep_stride);
  } else if (k == 3) {
    hipLaunchKernelGGL((k_dw_bwd_weight<3, 3>), dim3(gw), dim3(256), 0, s, *x, *y, *gy, dw, stride, pad, dil, 0, nrep, rep_stride);
  } else {
    for (int kh = 0; kh < 7; ++kh)
      hipLaunchKernelGGL((k_dw_bwd_weight<7, 1>), dim3(gw), dim3(256), 0, s, *x, *y, *gy, dw, stride, pad, dil, kh, nrep, rep_stride);
  }
  LHN_CHECK_LAUNCH("lhn_conv_dw_bwd");
  return 0;
}

extern "C" int lhn_conv_stem_bwd(const float* img, const lhn_view* y, const lhn_gradview* gy, float* dw, int Hi, int Wi, int k,
                                 int stride, int pad, int nrep, int64_t rep_stride, void* stream) {
  if (nrep < 1) nrep = 1;
  LHN_CHECK_ARG(img && lhn_view_ok(y) && gy && gy->dz && dw, "lhn_conv_stem_bwd: bad view / null pointer");
  LHN_CHECK_ARG(pow2(y->C / 4) && y->C <= 256 && (k == 1 || k == 3 || k == 7), "lhn_conv_stem_bwd: Cout=%d k=%d", y->C, k);
  const int PL = 256 / (y->C / 4);
  const int g = grid_for((int64_t)y->N * y->H * y->W, PL, 4);
  if (k == 7 && y->C == 64 && !lhn_dw_force_gather()) {
    const size_t lds = (size_t)64 * (68 + 161) * 4;
    static LhnKernelCfg cfg7;
    int per_cu = 1;
    if (!lhn_kernel_cfg(cfg7, &k_stem7_bwd_mfma, lds, 2, &per_cu)) {
      lhn_set_error("lhn_conv_stem_bwd: cannot reserve %zu B of LDS", lds);
      return 2;
    }
    const int64_t ntiles = ((int64_t)y->N * y->H * y->W + 63) / 64;
    int grid = lhn_num_cus() * per_cu;
    if (grid > ntiles) grid = (int)ntiles;
    hipLaunchKernelGGL(k_stem7_bwd_mfma, dim3(grid), dim3(256), lds, (hipStream_t)stream, img, *y, *gy, dw, Hi, Wi, stride, pad, nrep, rep_stride);
  } else if (k == 3 && y->C == 32 && !lhn_dw_force_gather()) {
    const size_t lds = (size_t)256 * (36 + 33) * 4;
    static LhnKernelCfg cfg;
    int per_cu = 1;
    if (!lhn_kernel_cfg(cfg, &k_stem3_bwd_mfma, lds, 2, &per_cu)) {
      lhn_set_error("lhn_conv_stem_bwd: cannot reserve %zu B of LDS", lds);
      return 2;
    }
    const int64_t ntiles = ((int64_t)y->N * y->H * y->W + 255) / 256;
    int grid = lhn_num_cus() * per_cu;
    if (grid > ntiles) grid = (int)ntiles;
    hipLaunchKernelGGL(k_stem3_bwd_mfma, dim3(grid), dim3(256), lds, (hipStream_t)stream, img, *y, *gy, dw, Hi, Wi, stride, pad, nrep, rep_stride);
  } else if (k == 3)
    hipLaunchKernelGGL((k_stem_bwd<3, 3>), dim3(g), dim3(256), 0, (hipStream_t)stream, img, *y, *gy, dw, Hi, Wi, stride, pad, nrep, rep_stride, 0);
  else if (k == 1)
    hipLaunchKernelGGL((k_stem_bwd<1, 1>), dim3(g), dim3(256), 0, (hipStream_t)stream, img, *y, *gy, dw, Hi, Wi, stride, pad, nrep, rep_stride, 0);
  else
    for (int kh = 0; kh < 7; ++kh)
      hipLaunchKernelGGL((k_stem_bwd<7, 1>), dim3(g), dim3(256), 0, (hipStream_t)stream, img, *y, *gy, dw, Hi, Wi, stride, pad, nrep, rep_stride, kh);
  LHN_CHECK_LAUNCH("lhn_conv_stem_bwd");
  return 0;
}

// =====================================================================================================
// LDS-staged K x K depthwise (stride 1, dilation DIL, pad DIL*(K-1)/2) over 32-channel groups.
// Block tile = TH x TW output pixels x 32 channels; the (TH+2P) x (TW+2P) input halo tile is loaded ONCE,
// transformed (pending BN / activation / gate) once, and parked in LDS; every thread (= 4 channels x one
// output column) walks down the tile rows (3x3/dil 1: sliding register window, 3 ds_read_b128 per output).
template <int K, int DIL, int TH, int TW>
struct DwTile {
  static constexpr int P = DIL * (K - 1) / 2, HH = TH + 2 * P, WW = TW + 2 * P, PIX = HH * WW;
};

template <int K, int DIL, int NS = 1>
__global__ void __launch_bounds__(256) k_dwk_fwd_lds(lhn_view x, const float* __restrict__ w, lhn_view y,
                                                     double* __restrict__ stats, int tiles_h, int tiles_w, int cgroups,
                                                     lhn_bnfin fin, int ps, DwExtra ex, int xchunk) {
  constexpr int TH = 8, TW = 32, KK = K * K;
  // XCD-aware tile order: blocks are dealt round-robin over the 8 XCDs, so logical block (b % 8) * xchunk + b / 8 gives each
  // XCD a CONTIGUOUS run of tiles -- neighbouring tiles (which share their halo rows / columns) then meet in one L2
  const int bid = xchunk ? (int)(blockIdx.x & 7) * xchunk + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  using T = DwTile<K, DIL, TH, TW>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f4* tile = reinterpret_cast<f4*>(smem);                  // [PIX][8] float4
  f4* red = tile + T::PIX * 8;                             // [256][2]
  f4* wl = red + 512;                                      // [KK][8] weights of this block's channel group
  const int tid = threadIdx.x, c4 = tid & 7, pl = tid >> 3;   // pl = output column 0..31
  // ps = pixel stride of the lattice a tile lives on.  ps == 2 runs a dilation-2 convolution as FOUR independent
  // dilation-1 convolutions on the parity sub-lattices (pixels (2i+a, 2j+b) only meet pixels of the same parity): the halo
  // shrinks from (TH+4)(TW+4) to (TH+2)(TW+2) and every pixel still is one contiguous 128-byte channel group.
  const int ntile = y.N * ps * ps * tiles_h * tiles_w * cgroups;
  const int cg = bid % cgroups;                            // grid % cgroups == 0 (host): fixed per block
  // C % 32 != 0 (lite_hrnet.py: 20 / 40 / 80 channels): the last group has cvalid < 8 float4 lanes; the others load the
  // group's first lane again (valid memory) and neither store nor count
  const int cvalid = min(8, (x.C - cg * 32) >> 2);
  const bool cok = c4 < cvalid;
  const int c4e = cok ? c4 : 0;
  const int cin = x.coff + cg * 32 + 4 * c4e, cout = y.coff + cg * 32 + 4 * c4e;
  const int cin2 = NS > 1 ? ex.v.coff + cg * 32 + 4 * c4e : 0;
  Xf4 xf, xf2;        // filled after the first tile's loads have been issued (pending BatchNorms are finalized meanwhile)
  for (int i = tid; i < KK * 8; i += 256) {
    const int k = i >> 3, cc = cg * 32 + 4 * min(i & 7, cvalid - 1);
    wl[i] = (f4){w[(cc + 0) * KK + k], w[(cc + 1) * KK + k], w[(cc + 2) * KK + k], w[(cc + 3) * KK + k]};
  }
  double sd[4] = {0, 0, 0, 0}, qd[4] = {0, 0, 0, 0};      // per-tile fp32 partials promoted to double (see k_conv_pw.hip)
  // halo staging in two phases: issue() sends every global load of a tile (clamped coordinates, no branches, so they
  // are all in flight together) into registers one tile AHEAD; commit() transforms, zeroes the padding and writes LDS
  constexpr int NIT = (T::PIX + 31) / 32;
  f4 raw[NIT], raw2[NS > 1 ? NIT : 1];
  auto issue = [&](int t) __attribute__((always_inline)) {
    int r = t / cgroups;
    const int tw = r % tiles_w;
    r /= tiles_w;
    const int th = r % tiles_h;
    r /= tiles_h;
    const int par = r % (ps * ps), n = r / (ps * ps), pa = par / ps, pb = par % ps;
    const int SH = (x.H - pa + ps - 1) / ps, SW = (x.W - pb + ps - 1) / ps;     // sub-lattice extent
    const int h0 = th * TH - T::P, w0 = tw * TW - T::P;
    const float* xin = x.data + (size_t)n * x.H * x.W * x.cstride + cin;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = min(pl + 32 * it, T::PIX - 1);
      const int ph = i / T::WW, pw = i - ph * T::WW;
      const int ih = pa + ps * min(max(h0 + ph, 0), max(SH - 1, 0)), iw = pb + ps * min(max(w0 + pw, 0), max(SW - 1, 0));
      const size_t pix = (size_t)min(ih, x.H - 1) * x.W + min(iw, x.W - 1);
      raw[it] = *reinterpret_cast<const f4*>(xin + pix * x.cstride);
    }
  };
  // second source: loaded at commit time, NOT a tile ahead (44 more live registers would drop the kernel to one block per
  // CU); the other resident block covers the latency
  auto issue2 = [&](int t) __attribute__((always_inline)) {
    int r = t / cgroups;
    const int tw = r % tiles_w;
    r /= tiles_w;
    const int th = r % tiles_h;
    r /= tiles_h;
    const int par = r % (ps * ps), n = r / (ps * ps), pa = par / ps, pb = par % ps;
    const int SH = (x.H - pa + ps - 1) / ps, SW = (x.W - pb + ps - 1) / ps;
    const int h0 = th * TH - T::P, w0 = tw * TW - T::P;
    const float* xin2 = ex.v.data + (size_t)n * x.H * x.W * ex.v.cstride + cin2;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = min(pl + 32 * it, T::PIX - 1);
      const int ph = i / T::WW, pw = i - ph * T::WW;
      const int ih = pa + ps * min(max(h0 + ph, 0), max(SH - 1, 0)), iw = pb + ps * min(max(w0 + pw, 0), max(SW - 1, 0));
      raw2[it] = *reinterpret_cast<const f4*>(xin2 + ((size_t)min(ih, x.H - 1) * x.W + min(iw, x.W - 1)) * ex.v.cstride);
    }
  };
  int t = bid;
  if (t < ntile) issue(t);
  // (the tile region is free until the first commit, which follows a barrier)
  xf = lhn_load_xf(x, cin);
  if (NS > 1) xf2 = lhn_load_xf(ex.v, cin2);
  for (; t < ntile; t += gridDim.x) {
    int r = t / cgroups;
    const int tw = r % tiles_w;
    r /= tiles_w;
    const int th = r % tiles_h;
    r /= tiles_h;
    const int par = r % (ps * ps), n = r / (ps * ps), pa = par / ps, pb = par % ps;
    const int SH = (x.H - pa + ps - 1) / ps, SW = (x.W - pb + ps - 1) / ps;
    f4 gate = x.gate ? *reinterpret_cast<const f4*>(x.gate + (size_t)n * x.cstride + cin) : (f4){1.f, 1.f, 1.f, 1.f};
    f4 gate2 = (f4){0.f, 0.f, 0.f, 0.f};
    if (NS > 1) {
      gate *= ex.coef[0];
      gate2 = (ex.v.gate ? *reinterpret_cast<const f4*>(ex.v.gate + (size_t)n * ex.v.cstride + cin2) : (f4){1.f, 1.f, 1.f, 1.f}) * ex.coef[1];
    }
    const int h0 = th * TH - T::P, w0 = tw * TW - T::P;
    if (NS > 1) issue2(t);
    __syncthreads();   // previous tile fully consumed (and wl visible)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = pl + 32 * it;
      if (i < T::PIX) {
        const int ph = i / T::WW, pw = i - ph * T::WW;
        const int ih = h0 + ph, iw = w0 + pw;
        const bool inb = ih >= 0 && ih < SH && iw >= 0 && iw < SW;
        f4 v = lhn_apply_xf(raw[it], xf) * gate;
        if (NS > 1) {
          v += lhn_apply_xf(raw2[it], xf2) * gate2;
          if (ex.sum_out && cok && inb && ph >= T::P && ph < T::P + TH && pw >= T::P && pw < T::P + TW)
            *reinterpret_cast<f4*>(ex.sum_out + ((size_t)(n * x.H + pa + ps * ih) * x.W + pb + ps * iw) * ex.so_cstride + ex.so_coff +
                                   cg * 32 + 4 * c4) = v;
        }
        tile[i * 8 + c4] = inb ? v : (f4){0.f, 0.f, 0.f, 0.f};
      }
    }
    __syncthreads();
    if (t + (int)gridDim.x < ntile) issue(t + gridDim.x);
    const int wo = tw * TW + pl;
    f4 s = (f4){0.f, 0.f, 0.f, 0.f}, q = s, kk = s;      // shifted by the tile's first value (see TileStat)
    int cnt = 0;
    if (wo < SW && cok) {
      const f4* col = tile + (pl + T::P) * 8 + c4;    // centre column of this thread, tile row 0
      float* yout = y.data + ((size_t)(n * y.H + pa + ps * th * TH) * y.W + pb + ps * wo) * y.cstride + cout;
      if (K == 3 && DIL == 1) {
        f4 wt[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) wt[k] = wl[k * 8 + c4];
        f4 win[3][3];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) win[a + 1][b] = col[(a * T::WW + (b - 1)) * 8];
#pragma unroll
        for (int rr = 0; rr < TH; ++rr) {
#pragma unroll
          for (int b = 0; b < 3; ++b) {
            win[0][b] = win[1][b];
            win[1][b] = win[2][b];
            win[2][b] = col[((rr + 2) * T::WW + (b - 1)) * 8];
          }
          if (th * TH + rr < SH) {
            f4 acc = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
              for (int b = 0; b < 3; ++b) acc += win[a][b] * wt[a * 3 + b];
            *reinterpret_cast<f4*>(yout + (size_t)rr * ps * y.W * y.cstride) = acc;
            if (cnt == 0) kk = acc;
            const f4 d = acc - kk;
            s += d;
            q += d * d;
            ++cnt;
          }
        }
      } else {
        for (int rr = 0; rr < TH; ++rr) {
          if (th * TH + rr >= SH) break;
          f4 acc = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int a = 0; a < K; ++a)
#pragma unroll
            for (int b = 0; b < K; ++b)
              acc += col[((rr + a * DIL) * T::WW + (b * DIL - T::P)) * 8] * wl[(a * K + b) * 8 + c4];
          *reinterpret_cast<f4*>(yout + (size_t)rr * ps * y.W * y.cstride) = acc;
          if (cnt == 0) kk = acc;
          const f4 d = acc - kk;
          s += d;
          q += d * d;
          ++cnt;
        }
      }
    }
    lhn_unshift4(sd, qd, s, q, kk, cnt);
  }
  if (stats) {
    const int C = x.C;
    double* st = stats + (size_t)((bid / cgroups) % LHN_STAT_REPLICAS) * 2 * C + cg * 32;
    lhn_block_stat_atomics_d(sd, qd, 8, reinterpret_cast<double*>(red), st, st + C, cvalid);
    if (fin.counter && lhn_last_block(fin.counter)) lhn_bn_finalize_block_par(fin, stats, reinterpret_cast<double*>(red));
  }
}

// Fused backward (dgrad + wgrad), TH x TW = 8 x 16: the dy halo tile (formed once per element from dz, raw y and
// the BN-backward coefficients) and the transformed x halo tile sit in LDS.
//   dx[h,w]   = sum_taps dy[h+P-a*DIL, w+P-b*DIL] * wgt[a][b]
//   dW[a][b] += sum_{h,w in tile} dy[h,w] * x[h-P+a*DIL, w-P+b*DIL]
// K = 3 keeps all 9 dW accumulators in registers; K = 7 keeps the per-tile partials in LDS (dws) instead.
// BNS: the input x is the output of a convolution + BatchNorm whose ONLY reader is this depthwise convolution (RepBasicUnit:
// 1x1 -> 3x3 depthwise, litehourglass.py:58-60).  The kernel then holds everything the producer's BatchNorm backward needs
// -- d(value of x) (the dx it just computed), the raw x, its table -- and accumulates sum(du), sum(du * xhat) per channel
// into the producer's replicated sums: the separate lhn_bn_bwd_reduce pass (re-reads x and dx: 134 MB, 20 us per unit at
// 64x64) disappears.
struct DwBnSum {
  double* sums;          // [LHN_STAT_REPLICAS][2][C] of the producer's BatchNorm backward, or NULL
  const float* save;     // [2][C] mean | invstd of the producer
  int C, coff;           // producer channels; channel of the producer that x's first channel is
  const float* add[2];   // gradient buffers (dx's geometry) whose values join the stored dx: d(sum) of residual adds that
                         // read x, so that no separate elementwise pass copies / accumulates them (NULL: none)
};

template <int K, int DIL, bool BNS = false>
__global__ void __launch_bounds__(256) k_dwk_bwd_lds(lhn_view x, const float* __restrict__ w, lhn_view y, lhn_gradview gy,
                                                     float* __restrict__ dx, int dx_acc, float* __restrict__ dw, int tiles_h,
                                                     int tiles_w, int cgroups, int nrep, int64_t rep_stride, int ps, DwBnSum bs, int xchunk) {
  constexpr int TH = 8, TW = 16, KK = K * K;
  const int bid = xchunk ? (int)(blockIdx.x & 7) * xchunk + (int)(blockIdx.x >> 3) : (int)blockIdx.x;     // see k_dwk_fwd_lds
  constexpr bool REGACC = (K == 3);
  using T = DwTile<K, DIL, TH, TW>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f4* tdy = reinterpret_cast<f4*>(smem);        // [PIX][8]
  f4* tx = tdy + T::PIX * 8;                    // [PIX][8]
  f4* red = tx + T::PIX * 8;                    // [256]
  f4* wl = red + 256;                           // [KK][8]
  f4* dws = wl + KK * 8;                        // [KK][8]  (K = 7 only) block-level dW partials
  f4* traw = dws + KK * 8;                      // [TH*TW][8] raw x of the tile interior (BNS only)
  f4* tmi = traw + TH * TW * 8;                 // [2][8] mean | invstd of this block's channel group (BNS only)
  const int tid = threadIdx.x, c4 = tid & 7, pl = tid >> 3;
  const int colw = pl & 15, rpar = pl >> 4;
  const int ntile = y.N * ps * ps * tiles_h * tiles_w * cgroups;      // ps: see k_dwk_fwd_lds
  const int cg = bid % cgroups;
  const int cvalid = min(8, (x.C - cg * 32) >> 2);          // see k_dwk_fwd_lds
  const bool cok = c4 < cvalid;
  const int c4e = cok ? c4 : 0;
  const int cx = x.coff + cg * 32 + 4 * c4e, cy = y.coff + cg * 32 + 4 * c4e;
  const Xf4 xxf = lhn_load_xf(x, cx), yxf = lhn_load_xf(y, cy);
  const Gr4 ygr = lhn_load_coef(gy, y.cstride, cy);
  for (int i = tid; i < KK * 8; i += 256) {
    const int k = i >> 3, cc = cg * 32 + 4 * min(i & 7, cvalid - 1);
    wl[i] = (f4){w[(cc + 0) * KK + k], w[(cc + 1) * KK + k], w[(cc + 2) * KK + k], w[(cc + 3) * KK + k]};
    if (!REGACC) dws[i] = (f4){0.f, 0.f, 0.f, 0.f};
  }
  f4 accw[REGACC ? KK : 1];
#pragma unroll
  for (int k = 0; k < (REGACC ? KK : 1); ++k) accw[k] = (f4){0.f, 0.f, 0.f, 0.f};
  f4 sdu = (f4){0.f, 0.f, 0.f, 0.f}, sdux = sdu;
  if (BNS && tid < 16) {
    const int kind = tid >> 3, cc = tid & 7;
    tmi[tid] = *reinterpret_cast<const f4*>(bs.save + kind * bs.C + bs.coff + cg * 32 + 4 * cc);
  }
  for (int t = bid; t < ntile; t += gridDim.x) {
    int r = t / cgroups;
    const int tw = r % tiles_w;
    r /= tiles_w;
    const int th = r % tiles_h;
    r /= tiles_h;
    const int par = r % (ps * ps), n = r / (ps * ps), pa = par / ps, pb = par % ps;
    const int SH = (x.H - pa + ps - 1) / ps, SW = (x.W - pb + ps - 1) / ps;
    const f4 xgate = x.gate ? *reinterpret_cast<const f4*>(x.gate + (size_t)n * x.cstride + cx) : (f4){1.f, 1.f, 1.f, 1.f};
    const f4 ygate = y.gate ? *reinterpret_cast<const f4*>(y.gate + (size_t)n * y.cstride + cy) : (f4){1.f, 1.f, 1.f, 1.f};
    const int h0 = th * TH - T::P, w0 = tw * TW - T::P;
    __syncthreads();
    {
      // all global loads of the halo tile first (clamped, branch-free), then the dy / x transforms and the LDS stores
      constexpr int NIT = (T::PIX + 31) / 32;
      f4 rx[NIT], ry[NIT], rz[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int i = min(pl + 32 * it, T::PIX - 1);
        const int ph = i / T::WW, pw = i - ph * T::WW;
        const int ih = min(pa + ps * min(max(h0 + ph, 0), max(SH - 1, 0)), x.H - 1);
        const int iw = min(pb + ps * min(max(w0 + pw, 0), max(SW - 1, 0)), x.W - 1);
        const size_t pix = (size_t)(n * x.H + ih) * x.W + iw;
        rx[it] = *reinterpret_cast<const f4*>(x.data + pix * x.cstride + cx);
        ry[it] = *reinterpret_cast<const f4*>(y.data + pix * y.cstride + cy);
        rz[it] = *reinterpret_cast<const f4*>(gy.dz + pix * y.cstride + cy);
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int i = pl + 32 * it;
        if (i < T::PIX) {
          const int ph = i / T::WW, pw = i - ph * T::WW;
          const int ih = h0 + ph, iw = w0 + pw;
          const bool inb = ih >= 0 && ih < SH && iw >= 0 && iw < SW;     // stride 1, "same" padding: x and y share geometry
          const f4 vx = lhn_apply_xf(rx[it], xxf) * xgate;
          f4 e = rz[it] * ygate;
          const f4 u = ry[it] * yxf.sc + yxf.sh;
          const f4 dl = (f4){u.x > 0.f ? 1.f : yxf.sl.x, u.y > 0.f ? 1.f : yxf.sl.y, u.z > 0.f ? 1.f : yxf.sl.z,
                             u.w > 0.f ? 1.f : yxf.sl.w};
          if (gy.dpool && inb) e += lhn_dpool_sum(gy, y, n, pa + ps * ih, pb + ps * iw, cy);
          const f4 vy = ygr.A * (e * dl) + ygr.B * ry[it] + ygr.Cc;
          const f4 z = (f4){0.f, 0.f, 0.f, 0.f};
          tx[i * 8 + c4] = inb ? vx : z;
          tdy[i * 8 + c4] = inb ? vy : z;
          if (BNS && ph >= T::P && ph < T::P + TH && pw >= T::P && pw < T::P + TW) traw[((ph - T::P) * TW + pw - T::P) * 8 + c4] = rx[it];
        }
      }
    }
    __syncthreads();
    const int wcol = tw * TW + colw;
    if (REGACC) {
#pragma unroll
      for (int j = 0; j < TH / 2; ++j) {
        const int rr = 2 * j + rpar, hh = th * TH + rr;
        if (wcol < SW && hh < SH) {
          const int centre = ((rr + T::P) * T::WW + colw + T::P) * 8 + c4;
          const f4 dyc = tdy[centre];
          f4 accx = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int a = 0; a < K; ++a)
#pragma unroll
            for (int b = 0; b < K; ++b) {
              const int off = ((a * DIL - T::P) * T::WW + (b * DIL - T::P)) * 8;
              accx += tdy[centre - off] * wl[(a * K + b) * 8 + c4];
              accw[REGACC ? a * K + b : 0] += dyc * tx[centre + off];
            }
          if (BNS) {          // accx is the complete d(value of x): this kernel is x's only reader (host-checked, dx_acc == 0)
            const f4 raw = traw[(rr * TW + colw) * 8 + c4];
            const f4 u = raw * xxf.sc + xxf.sh;
            const f4 du = accx * (f4){u.x > 0.f ? 1.f : xxf.sl.x, u.y > 0.f ? 1.f : xxf.sl.y, u.z > 0.f ? 1.f : xxf.sl.z, u.w > 0.f ? 1.f : xxf.sl.w};
            sdu += du;
            sdux += du * ((raw - tmi[c4]) * tmi[8 + c4]);
          }
          if (dx && cok) {
            // (issuing these loads for all four rows BEFORE the tap loops -- one memory latency per tile instead of four -- was
            // measured: 2 more spilled registers in the BNS instance and variant B 8.19 -> 8.29 ms per step; not kept)
            float* o = dx + ((size_t)(n * x.H + pa + ps * hh) * x.W + pb + ps * wcol) * x.cstride + cx;
            if (dx_acc) accx += *reinterpret_cast<const f4*>(o);
            if (bs.add[0]) accx += *reinterpret_cast<const f4*>(bs.add[0] + (o - dx));
            if (bs.add[1]) accx += *reinterpret_cast<const f4*>(bs.add[1] + (o - dx));
            *reinterpret_cast<f4*>(o) = accx;
          }
        }
      }
    } else {
      // dgrad
      if (dx)
        for (int j = 0; j < TH / 2; ++j) {
          const int rr = 2 * j + rpar, hh = th * TH + rr;
          if (wcol < SW && hh < SH && cok) {
            const int centre = ((rr + T::P) * T::WW + colw + T::P) * 8 + c4;
            f4 accx = (f4){0.f, 0.f, 0.f, 0.f};
            for (int a = 0; a < K; ++a)
#pragma unroll
              for (int b = 0; b < K; ++b)
                accx += tdy[centre - ((a * DIL - T::P) * T::WW + (b * DIL - T::P)) * 8] * wl[(a * K + b) * 8 + c4];
            float* o = dx + ((size_t)(n * x.H + pa + ps * hh) * x.W + pb + ps * wcol) * x.cstride + cx;
            if (dx_acc) accx += *reinterpret_cast<const f4*>(o);
            if (bs.add[0]) accx += *reinterpret_cast<const f4*>(bs.add[0] + (o - dx));
            if (bs.add[1]) accx += *reinterpret_cast<const f4*>(bs.add[1] + (o - dx));
            *reinterpret_cast<f4*>(o) = accx;
          }
        }
      // wgrad: one kernel row at a time, K register accumulators, cross-lane sum through `red`
      for (int a = 0; a < K; ++a) {
        f4 ar[K];
#pragma unroll
        for (int b = 0; b < K; ++b) ar[b] = (f4){0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < TH / 2; ++j) {
          const int rr = 2 * j + rpar, hh = th * TH + rr;
          if (wcol < SW && hh < SH) {
            const int centre = ((rr + T::P) * T::WW + colw + T::P) * 8 + c4;
            const f4 dyc = tdy[centre];
#pragma unroll
            for (int b = 0; b < K; ++b) ar[b] += dyc * tx[centre + ((a * DIL - T::P) * T::WW + (b * DIL - T::P)) * 8];
          }
        }
#pragma unroll
        for (int b = 0; b < K; ++b) {
          // reduce over the 32 pixel lanes of each channel lane: xor-shuffle across lanes 8,16,32 then LDS for the 4 waves
          f4 v = ar[b];
#pragma unroll
          for (int o = 8; o < 64; o <<= 1) {
            v.x += __shfl_xor(v.x, o, 64);
            v.y += __shfl_xor(v.y, o, 64);
            v.z += __shfl_xor(v.z, o, 64);
            v.w += __shfl_xor(v.w, o, 64);
          }
          if ((tid & 63) < 8) red[(tid >> 6) * 8 + c4 + 32 * b] = v;
        }
        __syncthreads();
        if (tid < 8 * K) {
          const int b = tid >> 3, cc = tid & 7;
          dws[(a * K + b) * 8 + cc] += red[cc + 32 * b] + red[8 + cc + 32 * b] + red[16 + cc + 32 * b] + red[24 + cc + 32 * b];
        }
        __syncthreads();
      }
    }
  }
  if (BNS) {           // BatchNorm-backward sums of the producer: same block reduction as the forward statistics
    double* st = bs.sums + (size_t)((bid / cgroups) % LHN_STAT_REPLICAS) * 2 * bs.C + bs.coff + cg * 32;
    lhn_block_stat_atomics(sdu, sdux, 8, red, st, st + bs.C);
  }
  // ---- flush dW
  float* dwr = dw + (size_t)((bid / cgroups) % nrep) * rep_stride;
  if (REGACC) {
    // lanes of a wave that share c4 (lane bits 3..5 differ) meet by xor-shuffles, the four waves through LDS (the dy tile
    // is dead by now), then one thread per (tap, c4) adds four floats
#pragma unroll
    for (int k = 0; k < KK; ++k)
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) {
        accw[k].x += __shfl_xor(accw[k].x, o, 64);
        accw[k].y += __shfl_xor(accw[k].y, o, 64);
        accw[k].z += __shfl_xor(accw[k].z, o, 64);
        accw[k].w += __shfl_xor(accw[k].w, o, 64);
      }
    __syncthreads();
    if ((tid & 63) < 8)
#pragma unroll
      for (int k = 0; k < KK; ++k) tdy[(k * 4 + (tid >> 6)) * 8 + c4] = accw[k];
    __syncthreads();
    if (tid < KK * 8) {
      const int k = tid >> 3, cc = tid & 7;
      const f4 sacc = tdy[(k * 4 + 0) * 8 + cc] + tdy[(k * 4 + 1) * 8 + cc] + tdy[(k * 4 + 2) * 8 + cc] + tdy[(k * 4 + 3) * 8 + cc];
      const int cb = cg * 32 + 4 * cc;
      if (cc < cvalid) {
        atomicAdd(dwr + (cb + 0) * KK + k, sacc.x);
        atomicAdd(dwr + (cb + 1) * KK + k, sacc.y);
        atomicAdd(dwr + (cb + 2) * KK + k, sacc.z);
        atomicAdd(dwr + (cb + 3) * KK + k, sacc.w);
      }
    }
  } else {
    __syncthreads();
    for (int i = tid; i < KK * 8; i += 256) {
      const int k = i >> 3, cb = cg * 32 + 4 * (i & 7);
      const f4 v = dws[i];
      if ((i & 7) >= cvalid) continue;
      atomicAdd(dwr + (cb + 0) * KK + k, v.x);
      atomicAdd(dwr + (cb + 1) * KK + k, v.y);
      atomicAdd(dwr + (cb + 2) * KK + k, v.z);
      atomicAdd(dwr + (cb + 3) * KK + k, v.w);
    }
  }
}

// blocks per XCD when the XCD-aware tile order applies (grid a multiple of 8 and of 8 * cgroups), else 0 = plain order.
// Measured on MI355X (variant B, bs64 256x256): giving each XCD a contiguous run of tiles is SLOWER than the plain
// round-robin order -- forward 3.08 vs 2.93 ms, train step 9.66 vs 9.55 ms: the halo re-reads already hit the Infinity
// Cache, while eight XCDs each streaming one contiguous region load the HBM channels less evenly.  Off unless LHN_XCD_ORDER=1.
static int dw3_xchunk(int grid, int cgroups) {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("LHN_XCD_ORDER");
    on = (e && e[0] == '1') ? 1 : 0;
  }
  return (on && grid % (8 * cgroups) == 0) ? grid / 8 : 0;
}
static int dw3_grid(int ntile, int cgroups, int per_cu) {
  int g = lhn_num_cus() * per_cu;
  g -= g % cgroups;
  if (g > ntile) g = ntile;       // ntile is a multiple of cgroups
  if (g < cgroups) g = cgroups;
  return g;
}

template <int K, int DIL, int NS = 1>
static void launch_dwk_fwd(const lhn_view* x, const float* w, const lhn_view* y, double* stats, lhn_bnfin fin, hipStream_t s, int ps = 1,
                           const DwExtra* exp = nullptr) {
  DwExtra ex;
  if (exp) ex = *exp; else { ex.n = 0; ex.sum_out = nullptr; }
  constexpr int TH = 8, TW = 32, P = DIL * (K - 1) / 2;
  const int cg = (x->C + 31) / 32;
  const int sh = (y->H + ps - 1) / ps, sw = (y->W + ps - 1) / ps;       // largest parity sub-lattice
  const int th = (sh + TH - 1) / TH, tw = (sw + TW - 1) / TW, ntile = y->N * ps * ps * th * tw * cg;
  const size_t lds = (size_t)((TH + 2 * P) * (TW + 2 * P) * 8 + 512 + K * K * 8) * 16;
  const int per_cu = lds > 80 * 1024 ? 1 : (lds > 52 * 1024 ? 2 : 3);
  static LhnKernelCfg cfg;
  (void)lhn_kernel_cfg(cfg, &k_dwk_fwd_lds<K, DIL, NS>, lds, 4, nullptr);
  const int grid = dw3_grid(ntile, cg, per_cu * 2);      // (2 / 3 / 4 / 8 / 16 blocks per CU measured in round 3: forward 2.65 / 2.68 / 2.66 / 2.71 / 2.71 ms)
  hipLaunchKernelGGL((k_dwk_fwd_lds<K, DIL, NS>), dim3(grid), dim3(256), lds, s, *x, w, *y, stats, th, tw, cg, fin, ps, ex, dw3_xchunk(grid, cg));
}
template <int K, int DIL, bool BNS = false>
static void launch_dwk_bwd(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, float* dx, int dx_acc,
                           float* dw, int nrep, int64_t rep_stride, hipStream_t s, int ps = 1, const DwBnSum* bsp = nullptr) {
  constexpr int TH = 8, TW = 16, P = DIL * (K - 1) / 2;
  DwBnSum bs;
  if (bsp) bs = *bsp; else { bs.sums = nullptr; bs.save = nullptr; bs.C = bs.coff = 0; bs.add[0] = bs.add[1] = nullptr; }
  const int cg = (x->C + 31) / 32;
  const int sh = (x->H + ps - 1) / ps, sw = (x->W + ps - 1) / ps;
  const int th = (sh + TH - 1) / TH, tw = (sw + TW - 1) / TW, ntile = x->N * ps * ps * th * tw * cg;
  const size_t lds = (size_t)((TH + 2 * P) * (TW + 2 * P) * 16 + 256 + 2 * K * K * 8 + (BNS ? TH * TW * 8 + 16 : 0)) * 16;
  static LhnKernelCfg cfg;
  (void)lhn_kernel_cfg(cfg, &k_dwk_bwd_lds<K, DIL, BNS>, lds, 4, nullptr);
  const int grid = dw3_grid(ntile, cg, 4);
  hipLaunchKernelGGL((k_dwk_bwd_lds<K, DIL, BNS>), dim3(grid), dim3(256), lds, s, *x, w, *y, *gy, dx, dx_acc, dw, th, tw,
                     cg, nrep, rep_stride, ps, bs, dw3_xchunk(grid, cg));
}


// =====================================================================================================
// Stride-2 3x3 depthwise (pad 1) with the input tile in LDS: the downsampling convolutions of the stems and of
// lite_hrnet.py's fuse / transition layers (54 per Lite-HRNet-18 step).  The row-gather kernels they used to take
// (k_dw_fwd / k_dw_bwd_data + k_dw_bwd_weight) ran at a fifth of the traffic-bound time.
// Output tile TH x TW = 4 x 16, 32 channels per block; input tile (2 TH + 1) x (2 TW + 1) with origin (2 oh0 - 1, 2 ow0 - 1).
struct DwS2 {
  static constexpr int TH = 4, TW = 16, XH = 2 * TH + 1, XW = 2 * TW + 1, XPIX = XH * XW, DH = TH + 1, DW = TW + 1, DPIX = DH * DW;
};

__global__ void __launch_bounds__(256) k_dws2_fwd_lds(lhn_view x, const float* __restrict__ w, lhn_view y, double* __restrict__ stats,
                                                      int tiles_h, int tiles_w, int cgroups, lhn_bnfin fin) {
  using T = DwS2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f4* tx = reinterpret_cast<f4*>(smem);          // [XPIX][8]
  f4* red = tx + T::XPIX * 8;                    // [512]
  f4* wl = red + 512;                            // [9][8]
  const int tid = threadIdx.x, c4 = tid & 7, pl = tid >> 3;
  const int ntile = y.N * tiles_h * tiles_w * cgroups;
  const int cg = blockIdx.x % cgroups;
  const int cvalid = min(8, (x.C - cg * 32) >> 2);          // see k_dwk_fwd_lds
  const bool cok = c4 < cvalid;
  const int c4e = cok ? c4 : 0;
  const int cin = x.coff + cg * 32 + 4 * c4e, cout = y.coff + cg * 32 + 4 * c4e;
  for (int i = tid; i < 72; i += 256) {
    const int k = i >> 3, cc = cg * 32 + 4 * min(i & 7, cvalid - 1);
    wl[i] = (f4){w[(cc + 0) * 9 + k], w[(cc + 1) * 9 + k], w[(cc + 2) * 9 + k], w[(cc + 3) * 9 + k]};
  }
  const Xf4 xf = lhn_load_xf(x, cin);
  double sd[4] = {0, 0, 0, 0}, qd[4] = {0, 0, 0, 0};
  constexpr int NIT = (T::XPIX + 31) / 32;
  for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
    int r = t / cgroups;
    const int tw = r % tiles_w;
    r /= tiles_w;
    const int th = r % tiles_h, n = r / tiles_h;
    const int h0 = 2 * th * T::TH - 1, w0 = 2 * tw * T::TW - 1;
    const f4 gate = x.gate ? *reinterpret_cast<const f4*>(x.gate + (size_t)n * x.cstride + cin) : (f4){1.f, 1.f, 1.f, 1.f};
    const float* xin = x.data + (size_t)n * x.H * x.W * x.cstride + cin;
    f4 raw[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = min(pl + 32 * it, T::XPIX - 1);
      const int ph = i / T::XW, pw = i - ph * T::XW;
      const int ih = min(max(h0 + ph, 0), x.H - 1), iw = min(max(w0 + pw, 0), x.W - 1);
      raw[it] = *reinterpret_cast<const f4*>(xin + ((size_t)ih * x.W + iw) * x.cstride);
    }
    __syncthreads();                 // previous tile consumed (and wl / the resolved table visible)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = pl + 32 * it;
      if (i < T::XPIX) {
        const int ph = i / T::XW, pw = i - ph * T::XW;
        const int ih = h0 + ph, iw = w0 + pw;
        const bool inb = ih >= 0 && ih < x.H && iw >= 0 && iw < x.W;
        tx[i * 8 + c4] = inb ? lhn_apply_xf(raw[it], xf) * gate : (f4){0.f, 0.f, 0.f, 0.f};
      }
    }
    __syncthreads();
    f4 s = (f4){0.f, 0.f, 0.f, 0.f}, q = s, kk = s;
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int o = pl + 32 * j, oh = o / T::TW, ow = o - oh * T::TW;
      const int ho = th * T::TH + oh, wo = tw * T::TW + ow;
      if (cok && ho < y.H && wo < y.W) {
        f4 acc = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) acc += tx[((2 * oh + a) * T::XW + 2 * ow + b) * 8 + c4] * wl[(a * 3 + b) * 8 + c4];
        *reinterpret_cast<f4*>(y.data + ((size_t)(n * y.H + ho) * y.W + wo) * y.cstride + cout) = acc;
        if (cnt == 0) kk = acc;
        const f4 d = acc - kk;
        s += d;
        q += d * d;
        ++cnt;
      }
    }
    lhn_unshift4(sd, qd, s, q, kk, cnt);
  }
  if (stats) {
    const int C = x.C;
    double* st = stats + (size_t)((blockIdx.x / cgroups) % LHN_STAT_REPLICAS) * 2 * C + cg * 32;
    lhn_block_stat_atomics_d(sd, qd, 8, reinterpret_cast<double*>(red), st, st + C, cvalid);
    if (fin.counter && lhn_last_block(fin.counter)) lhn_bn_finalize_block(fin, stats);
  }
}

// fused backward: dx for the 2TH x 2TW input pixels the tile owns (every input pixel belongs to one tile) and dW.
//   dx[ih,iw] = sum over taps (a,b) with (ih+1-a), (iw+1-b) even of dy[(ih+1-a)/2, (iw+1-b)/2] * w[a][b]
//   dW[a][b] += sum over the tile's outputs dy[ho,wo] * x[2ho-1+a, 2wo-1+b]
__global__ void __launch_bounds__(256, 2) k_dws2_bwd_lds(lhn_view x, const float* __restrict__ w, lhn_view y, lhn_gradview gy,
                                                      float* __restrict__ dx, int dx_acc, float* __restrict__ dw, int tiles_h,
                                                      int tiles_w, int cgroups, int nrep, int64_t rep_stride) {
  using T = DwS2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f4* tx = reinterpret_cast<f4*>(smem);          // [XPIX][8] transformed x (zero outside the image)
  f4* tdy = tx + T::XPIX * 8;                    // [DPIX][8] dy (zero outside the output map); >= 36*8 float4 for the flush
  f4* wl = tdy + T::DPIX * 8;                    // [9][8]
  const int tid = threadIdx.x, c4 = tid & 7, pl = tid >> 3;
  const int ntile = y.N * tiles_h * tiles_w * cgroups;
  const int cg = blockIdx.x % cgroups;
  const int cvalid = min(8, (x.C - cg * 32) >> 2);
  const bool cok = c4 < cvalid;
  const int c4e = cok ? c4 : 0;
  const int cx = x.coff + cg * 32 + 4 * c4e, cy = y.coff + cg * 32 + 4 * c4e;
  const Xf4 xxf = lhn_load_xf(x, cx), yxf = lhn_load_xf(y, cy);
  const Gr4 ygr = lhn_load_coef(gy, y.cstride, cy);
  for (int i = tid; i < 72; i += 256) {
    const int k = i >> 3, cc = cg * 32 + 4 * min(i & 7, cvalid - 1);
    wl[i] = (f4){w[(cc + 0) * 9 + k], w[(cc + 1) * 9 + k], w[(cc + 2) * 9 + k], w[(cc + 3) * 9 + k]};
  }
  f4 accw[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) accw[k] = (f4){0.f, 0.f, 0.f, 0.f};
  constexpr int NITX = (T::XPIX + 31) / 32, NITD = (T::DPIX + 31) / 32;
  for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
    int r = t / cgroups;
    const int tw = r % tiles_w;
    r /= tiles_w;
    const int th = r % tiles_h, n = r / tiles_h;
    const int h0 = 2 * th * T::TH - 1, w0 = 2 * tw * T::TW - 1;       // input tile origin
    const int oh0 = th * T::TH, ow0 = tw * T::TW;                     // dy tile origin
    const f4 xgate = x.gate ? *reinterpret_cast<const f4*>(x.gate + (size_t)n * x.cstride + cx) : (f4){1.f, 1.f, 1.f, 1.f};
    f4 rx[NITX], ry[NITD], rz[NITD];
#pragma unroll
    for (int it = 0; it < NITX; ++it) {
      const int i = min(pl + 32 * it, T::XPIX - 1);
      const int ph = i / T::XW, pw = i - ph * T::XW;
      const int ih = min(max(h0 + ph, 0), x.H - 1), iw = min(max(w0 + pw, 0), x.W - 1);
      rx[it] = *reinterpret_cast<const f4*>(x.data + ((size_t)(n * x.H + ih) * x.W + iw) * x.cstride + cx);
    }
#pragma unroll
    for (int it = 0; it < NITD; ++it) {
      const int i = min(pl + 32 * it, T::DPIX - 1);
      const int ph = i / T::DW, pw = i - ph * T::DW;
      const int ho = min(oh0 + ph, y.H - 1), wo = min(ow0 + pw, y.W - 1);
      const size_t off = ((size_t)(n * y.H + ho) * y.W + wo) * y.cstride + cy;
      ry[it] = *reinterpret_cast<const f4*>(y.data + off);
      rz[it] = *reinterpret_cast<const f4*>(gy.dz + off);
    }
    __syncthreads();                 // previous tile consumed (and wl visible)
#pragma unroll
    for (int it = 0; it < NITX; ++it) {
      const int i = pl + 32 * it;
      if (i < T::XPIX) {
        const int ph = i / T::XW, pw = i - ph * T::XW;
        const int ih = h0 + ph, iw = w0 + pw;
        const bool inb = ih >= 0 && ih < x.H && iw >= 0 && iw < x.W;
        tx[i * 8 + c4] = inb ? lhn_apply_xf(rx[it], xxf) * xgate : (f4){0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int it = 0; it < NITD; ++it) {
      const int i = pl + 32 * it;
      if (i < T::DPIX) {
        const int ph = i / T::DW, pw = i - ph * T::DW;
        const int ho = oh0 + ph, wo = ow0 + pw;
        f4 d = (f4){0.f, 0.f, 0.f, 0.f};
        if (ho < y.H && wo < y.W) {
          const f4 du = lhn_grad_du(y, gy, yxf, ry[it], rz[it], n, ho, wo, cy);
          d = ygr.A * du + ygr.B * ry[it] + ygr.Cc;
        }
        tdy[i * 8 + c4] = d;
      }
    }
    __syncthreads();
    // ---- dW: two outputs per thread
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int o = pl + 32 * j, oh = o / T::TW, ow = o - oh * T::TW;
      const f4 dyc = tdy[(oh * T::DW + ow) * 8 + c4];          // zero outside the map
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) accw[a * 3 + b] += dyc * tx[((2 * oh + a) * T::XW + 2 * ow + b) * 8 + c4];
    }
    // ---- dx: thread = input column c = pl of the tile (parity fixed), 2 TH rows
    if (dx && cok) {
      const int c = pl, iw = 2 * ow0 + c;
      // column taps: c even -> b = 1 at dy column c/2; c odd -> b = 0 at (c+1)/2 and b = 2 at (c-1)/2
      const int nb = (c & 1) ? 2 : 1;
      const int bA = (c & 1) ? 0 : 1, colA = (c & 1) ? (c + 1) >> 1 : c >> 1, colB = (c - 1) >> 1;
      if (iw < x.W) {
#pragma unroll
        for (int rr = 0; rr < 2 * T::TH; ++rr) {
          const int ih = 2 * oh0 + rr;
          if (ih < x.H) {
            f4 accx = (f4){0.f, 0.f, 0.f, 0.f};
            // row taps: rr even -> a = 1 at dy row rr/2; rr odd -> a = 0 at (rr+1)/2 and a = 2 at (rr-1)/2  (rr is a compile-time constant)
            if ((rr & 1) == 0) {
              const f4* drow = tdy + ((rr >> 1) * T::DW) * 8 + c4;
              accx += drow[colA * 8] * wl[(3 + bA) * 8 + c4];
              if (nb == 2) accx += drow[colB * 8] * wl[(3 + 2) * 8 + c4];
            } else {
              const f4* d0 = tdy + (((rr + 1) >> 1) * T::DW) * 8 + c4;
              const f4* d2 = tdy + (((rr - 1) >> 1) * T::DW) * 8 + c4;
              accx += d0[colA * 8] * wl[(0 + bA) * 8 + c4] + d2[colA * 8] * wl[(6 + bA) * 8 + c4];
              if (nb == 2) accx += d0[colB * 8] * wl[(0 + 2) * 8 + c4] + d2[colB * 8] * wl[(6 + 2) * 8 + c4];
            }
            float* o = dx + ((size_t)(n * x.H + ih) * x.W + iw) * x.cstride + cx;
            if (dx_acc) accx += *reinterpret_cast<const f4*>(o);
            *reinterpret_cast<f4*>(o) = accx;
          }
        }
      }
    }
  }
  // ---- flush dW (as k_dwk_bwd_lds)
  float* dwr = dw + (size_t)((blockIdx.x / cgroups) % nrep) * rep_stride;
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
      accw[k].x += __shfl_xor(accw[k].x, o, 64);
      accw[k].y += __shfl_xor(accw[k].y, o, 64);
      accw[k].z += __shfl_xor(accw[k].z, o, 64);
      accw[k].w += __shfl_xor(accw[k].w, o, 64);
    }
  __syncthreads();
  if ((tid & 63) < 8)
#pragma unroll
    for (int k = 0; k < 9; ++k) tdy[(k * 4 + (tid >> 6)) * 8 + c4] = accw[k];
  __syncthreads();
  if (tid < 72) {
    const int k = tid >> 3, cc = tid & 7;
    const f4 sacc = tdy[(k * 4 + 0) * 8 + cc] + tdy[(k * 4 + 1) * 8 + cc] + tdy[(k * 4 + 2) * 8 + cc] + tdy[(k * 4 + 3) * 8 + cc];
    const int cb = cg * 32 + 4 * cc;
    if (cc < cvalid) {
      atomicAdd(dwr + (cb + 0) * 9 + k, sacc.x);
      atomicAdd(dwr + (cb + 1) * 9 + k, sacc.y);
      atomicAdd(dwr + (cb + 2) * 9 + k, sacc.z);
      atomicAdd(dwr + (cb + 3) * 9 + k, sacc.w);
    }
  }
}

// returns 1 if the stride-2 tiled kernel was launched
static int dws2_fwd(const lhn_view* x, const float* w, const lhn_view* y, double* stats, lhn_bnfin fin, hipStream_t s) {
  using T = DwS2;
  const int cg = (x->C + 31) / 32;
  const int th = (y->H + T::TH - 1) / T::TH, tw = (y->W + T::TW - 1) / T::TW, ntile = y->N * th * tw * cg;
  size_t lds = (size_t)(T::XPIX * 8 + 512 + 72) * 16;
  static LhnKernelCfg cfg;
  if (!lhn_kernel_cfg(cfg, &k_dws2_fwd_lds, lds, 4, nullptr)) return 0;
  hipLaunchKernelGGL(k_dws2_fwd_lds, dim3(dw3_grid(ntile, cg, 6)), dim3(256), lds, s, *x, w, *y, stats, th, tw, cg, fin);
  return 1;
}
static int dws2_bwd(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, float* dx, int dx_acc, float* dw,
                    int nrep, int64_t rep_stride, hipStream_t s) {
  using T = DwS2;
  const int cg = (x->C + 31) / 32;
  const int th = (y->H + T::TH - 1) / T::TH, tw = (y->W + T::TW - 1) / T::TW, ntile = y->N * th * tw * cg;
  const size_t lds = (size_t)((T::XPIX + T::DPIX) * 8 + 72) * 16;
  static LhnKernelCfg cfg;
  if (!lhn_kernel_cfg(cfg, &k_dws2_bwd_lds, lds, 4, nullptr)) return 0;
  hipLaunchKernelGGL(k_dws2_bwd_lds, dim3(dw3_grid(ntile, cg, 4)), dim3(256), lds, s, *x, w, *y, *gy, dx, dx_acc, dw, th, tw, cg, nrep, rep_stride);
  return 1;
}

// returns 1 if an LDS-tiled kernel was launched
int lhn_dwk_fwd_lds(const lhn_view* x, const float* w, const lhn_view* y, double* stats, int k, int dil, lhn_bnfin fin,
                    hipStream_t s, const DwExtra* ex) {
  if (ex && ex->n > 0) {     // two summed sources: 3x3, dilation 1 or 2 (parity sub-lattices) -- MSRB's second round
    if (k == 3 && dil == 1) launch_dwk_fwd<3, 1, 2>(x, w, y, stats, fin, s, 1, ex);
    else if (k == 3 && dil == 2 && y->W >= 16) launch_dwk_fwd<3, 1, 2>(x, w, y, stats, fin, s, 2, ex);
    else return 0;
    return 1;
  }
  if (k == 3 && dil == 1) launch_dwk_fwd<3, 1>(x, w, y, stats, fin, s, 1, ex);
  else if (k == 3 && dil == 2 && y->W >= 16) launch_dwk_fwd<3, 1>(x, w, y, stats, fin, s, 2, ex);    // parity sub-lattices
  else if (k == 3 && dil == 2) launch_dwk_fwd<3, 2>(x, w, y, stats, fin, s, 1, ex);
  else if (k == 7 && dil == 1) launch_dwk_fwd<7, 1>(x, w, y, stats, fin, s, 1, ex);
  else return 0;
  return 1;
}
int lhn_dwk_bwd_lds(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, float* dx, int dx_acc,
                    float* dw, int k, int dil, int nrep, int64_t rep_stride, hipStream_t s, const DwBnSum* bs) {
  if (bs && bs->sums) {        // fused BatchNorm-backward sums of the producer: the 3x3 / dilation 1 instance only
    if (!(k == 3 && dil == 1 && dx && !dx_acc)) return 0;
    launch_dwk_bwd<3, 1, true>(x, w, y, gy, dx, dx_acc, dw, nrep, rep_stride, s, 1, bs);
    return 1;
  }
  if (k == 3 && dil == 1) launch_dwk_bwd<3, 1>(x, w, y, gy, dx, dx_acc, dw, nrep, rep_stride, s, 1, bs);
  else if (k == 3 && dil == 2 && x->W >= 16) launch_dwk_bwd<3, 1>(x, w, y, gy, dx, dx_acc, dw, nrep, rep_stride, s, 2, bs);
  else if (k == 3 && dil == 2) launch_dwk_bwd<3, 2>(x, w, y, gy, dx, dx_acc, dw, nrep, rep_stride, s, 1, bs);
  else if (k == 7 && dil == 1) launch_dwk_bwd<7, 1>(x, w, y, gy, dx, dx_acc, dw, nrep, rep_stride, s, 1, bs);
  else return 0;
  return 1;
}

static int dw_fwd_extra(const lhn_view* x, const float* w, const lhn_view* y, double* stats, int k, int stride, int pad, int dil,
                        lhn_bnfin fin, const lhn_view* extra, const float* coef, const lhn_view* so, hipStream_t s) {
  if (!(stride == 1 && pad == dil * (k - 1) / 2 && x->C % 4 == 0 && y->W >= 8)) return 0;
  if (!lhn_no_pend(x) || !lhn_no_pend(extra)) return 0;
  DwExtra ex;
  ex.v = *extra;
  ex.coef[0] = coef[0];
  ex.coef[1] = coef[1];
  ex.n = 1;
  ex.sum_out = so ? so->data : nullptr;
  ex.so_cstride = so ? so->cstride : 0;
  ex.so_coff = so ? so->coff : 0;
  return lhn_dwk_fwd_lds(x, w, y, stats, k, dil, fin, s, &ex);
}

static int dw_bwd_addends(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, float* dx, int dx_acc, float* dw,
                          int k, int dil, int nrep, int64_t rep_stride, const float* a0, const float* a1, hipStream_t s) {
  DwBnSum bs;
  bs.sums = nullptr;
  bs.save = nullptr;
  bs.C = bs.coff = 0;
  bs.add[0] = a0;
  bs.add[1] = a1;
  return lhn_dwk_bwd_lds(x, w, y, gy, dx, dx_acc, dw, k, dil, nrep, rep_stride, s, &bs);
}

static int dw_bwd_fused(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, float* dx, float* dw, int k,
                        int dil, int nrep, int64_t rep_stride, double* bn_sums, const float* bn_save, int bn_C, int bn_coff, hipStream_t s) {
  DwBnSum bs;
  bs.sums = bn_sums;
  bs.save = bn_save;
  bs.C = bn_C;
  bs.coff = bn_coff;
  bs.add[0] = bs.add[1] = nullptr;
  return lhn_dwk_bwd_lds(x, w, y, gy, dx, 0, dw, k, dil, nrep, rep_stride, s, &bs);
}
