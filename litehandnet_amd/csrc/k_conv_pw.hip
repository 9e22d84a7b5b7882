// 1x1 (pointwise) convolution on fp32 MFMA, NHWC, with the producer's pending BatchNorm/activation
// applied on load and this layer's BatchNorm statistics accumulated in the epilogue.
//   Y[m][co] = sum_ci X[m][ci] * W[co][ci]        m = output pixel, X = consumed value of the input view
// Tile: 128 pixels x (32*NT) output channels per 256-thread block; wave w owns pixel rows [32w,32w+32).
// MFMA: v_mfma_f32_32x32x2_f32 (exact fp32 fma chain).  Operands come from LDS images with a +4 float
// row pad so that every ds_read_b128 is conflict-free (row stride = odd number of 16-B slots).
// K order inside an 8-wide chunk is permuted (lane half h supplies k = 8*kc + 4*h + j at step j); both
// operands use the same permutation, so only the fp32 summation order differs from a sequential loop.
#include <stdlib.h>
#include "lhn_common.h"

// Runtime geometry of ONE launch.  Wide or odd channel counts (hourglass C = 256, lite-hrnet 40/80/160/320) are run by the
// host as a grid of (<= 128 input channels) x (<= 128 output channels) launches over channel slices of the same buffers:
//   * the weight tensor keeps its own row stride (`wstride` floats), so a slice is just a pointer offset;
//   * `kvalid` input channels of the slice are real, the rest of the CIN-wide tile is zero (in A and in W);
//   * rows >= `wrows` of the slice have zero weights and zero bias (a 21-feature head stored in a 24-channel buffer);
//   * `yacc`: this launch ADDS to what the previous input slice stored (K split); bias and BatchNorm statistics belong to
//     the last slice, which sees the complete sum.
struct PwGeom {
  int wstride, kvalid, wrows, yacc, statC;
  int64_t nchw_bstride;      // floats between two images of the NCHW output
};

// stage a [rows][CIN] weight slice into LDS (row pad LDA): zero beyond wrows / kvalid; 16-byte loads when the slice allows
template <int CIN>
__device__ __forceinline__ void pw_stage_w(float* Ws, int LDA, const float* __restrict__ w, int rows_pad, const PwGeom& g) {
  constexpr int C4 = CIN / 4;
  const bool vec = (g.wstride % 4 == 0) && (g.kvalid % 4 == 0) && ((reinterpret_cast<uintptr_t>(w) & 15) == 0);
  for (int i = threadIdx.x; i < rows_pad * C4; i += 256) {
    const int co = i / C4, k4 = i % C4;
    f4 v = (f4){0.f, 0.f, 0.f, 0.f};
    if (co < g.wrows) {
      const float* r = w + (int64_t)co * g.wstride + k4 * 4;
      if (vec) {
        if (4 * k4 < g.kvalid) v = *reinterpret_cast<const f4*>(r);
      } else {
        if (4 * k4 + 0 < g.kvalid) v.x = r[0];
        if (4 * k4 + 1 < g.kvalid) v.y = r[1];
        if (4 * k4 + 2 < g.kvalid) v.z = r[2];
        if (4 * k4 + 3 < g.kvalid) v.w = r[3];
      }
    }
    *reinterpret_cast<f4*>(Ws + co * LDA + k4 * 4) = v;
  }
}

// Extra input sources: the consumed input is  sum_s coef[s] * value_s  (value = gate * act(BN(raw)) of source s; source 0 is
// `x`).  This is how the residual sums of MSRB (litehourglass.py:41-49: out + ca(cat), then out + x) reach the 1x1 without
// ever being written to HBM: summed on load.  All sources share x's geometry and channel range width.
struct PwExtra {
  lhn_view v[2];
  float coef[3];
  int n;             // number of EXTRA sources in v (0..2)
  float* sum_out;    // the summed input is ALSO written here (NULL: not): training needs it once, for the weight gradient
  int so_cstride, so_coff;
};

template <int CIN, int NT, int NS = 1>
__global__ void __launch_bounds__(256, NS > 1 ? 1 : ((CIN == 64 && NT == 2) || (CIN == 32 && NT == 4) ? 3 : (CIN == 64 && NT == 4) ? 2 : 1)) k_pw_fwd(lhn_view x, const float* __restrict__ w, const float* __restrict__ bias,
                                                lhn_view y, double* __restrict__ stats, int stride,
                                                float* __restrict__ y_nchw, int cout, int M, int ntiles, lhn_bnfin fin,
                                                PwGeom geo, PwExtra ex) {
  constexpr int LDA = CIN + 4;
  constexpr int PF = CIN / 8;   // float4 loads per thread per 128-pixel tile
  constexpr int C4 = CIN / 4;   // float4 per pixel row
  constexpr int RP = 256 / C4;  // rows covered per pass
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                    // [32*NT][LDA]
  float* As = smem + 32 * NT * LDA;    // [128][LDA]
  float* red = As;                     // [4][32*NT][2]: aliases the A tile after the last tile's trailing barrier
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: wave-uniform predicates become scalar branches)
  const int l31 = lane & 31, lh = lane >> 5;

  const int c4 = tid % C4, row0 = tid / C4;
  const bool kok = 4 * c4 < x.C;                     // channel groups beyond the view are zero columns of the tile
  const int cabs = x.coff + (kok ? 4 * c4 : 0);
  const Xf4 xf = lhn_load_xf(x, cabs);
  const int HoWo = y.H * y.W;
  // extra sources (NS > 1): own buffer, table, gate, channel offset
  int ecabs[NS > 1 ? NS - 1 : 1];
  Xf4 exf[NS > 1 ? NS - 1 : 1];
  f4 epre[NS > 1 ? NS - 1 : 1][PF];
  if (NS > 1) {
#pragma unroll
    for (int e = 0; e < NS - 1; ++e)
      if (e < ex.n) {
        ecabs[e] = ex.v[e].coff + (kok ? 4 * c4 : 0);
        exf[e] = lhn_load_xf(ex.v[e], ecabs[e]);
      }
  }
  pw_stage_w<CIN>(Ws, LDA, w, 32 * NT, geo);

  f4 pre[PF];
  auto in_pix = [&](int m) __attribute__((always_inline)) -> int64_t {
    if (stride == 1) return m;
    const int n = m / HoWo, r = m - n * HoWo;
    const int ho = r / y.W, wo = r - ho * y.W;
    return ((int64_t)n * x.H + ho * stride) * x.W + wo * stride;
  };
  auto issue = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const int m = tile * 128 + row0 + p * RP;
      pre[p] = (f4){0.f, 0.f, 0.f, 0.f};
      if (m < M && kok) pre[p] = *reinterpret_cast<const f4*>(x.data + in_pix(m) * x.cstride + cabs);
      if (NS > 1) {
#pragma unroll
        for (int e = 0; e < NS - 1; ++e)
          if (e < ex.n && m < M && kok)
            epre[e][p] = *reinterpret_cast<const f4*>(ex.v[e].data + (int64_t)m * ex.v[e].cstride + ecabs[e]);
      }
    }
  };
  auto commit = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const int row = row0 + p * RP, m = tile * 128 + row;
      f4 v = (f4){0.f, 0.f, 0.f, 0.f};
      if (m < M && kok) {
        v = lhn_apply_xf(pre[p], xf);
        if (x.gate) v *= *reinterpret_cast<const f4*>(x.gate + (int64_t)(m / HoWo) * x.cstride + cabs);
        if (NS > 1) {
          v *= ex.coef[0];
#pragma unroll
          for (int e = 0; e < NS - 1; ++e)
            if (e < ex.n) {
              f4 u = lhn_apply_xf(epre[e][p], exf[e]);
              if (ex.v[e].gate) u *= *reinterpret_cast<const f4*>(ex.v[e].gate + (int64_t)(m / HoWo) * ex.v[e].cstride + ecabs[e]);
              v += u * ex.coef[e + 1];
            }
          if (ex.sum_out) *reinterpret_cast<f4*>(ex.sum_out + (int64_t)m * ex.so_cstride + ex.so_coff + 4 * c4) = v;
        }
      }
      *reinterpret_cast<f4*>(As + row * LDA + 4 * c4) = v;
    }
  };

  // BatchNorm statistics: shifted fp32 partial sums over ONE tile (16 values per lane), un-shifted into double per tile
  // (TileStat) -- a running fp32 sum of squares loses the variance when |mean| >> sigma (E[x^2] - E[x]^2 cancels)
  double ssum[NT], ssq[NT];
  float bv[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    ssum[nt] = ssq[nt] = 0.0;
    const int ch = nt * 32 + l31;
    bv[nt] = (bias && ch < geo.wrows) ? bias[ch] : 0.f;
  }

  int tile = blockIdx.x;
  if (tile < ntiles) issue(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    commit(tile);
    __syncthreads();
    const int next = tile + gridDim.x;
    if (next < ntiles) issue(next);

    f16v acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

    const float* arow = As + (wave * 32 + l31) * LDA + 4 * lh;
    const float* brow = Ws + l31 * LDA + 4 * lh;
#pragma unroll 4
    for (int kc = 0; kc < CIN / 8; ++kc) {
      const f4 a = *reinterpret_cast<const f4*>(arow + kc * 8);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f4 b = *reinterpret_cast<const f4*>(brow + nt * 32 * LDA + kc * 8);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[nt], 0, 0, 0);
      }
    }

    // epilogue: C/D layout col = lane&31 (channel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel)
    const int mbase = tile * 128 + wave * 32 + 4 * lh;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int ch = nt * 32 + l31;
      if (ch >= cout) continue;
      if (y_nchw) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int m0 = mbase + 8 * g;
          if (m0 < M) {  // HoWo % 4 == 0 is checked on the host: 4 consecutive pixels share an image
            const int n = m0 / HoWo, p = m0 - n * HoWo;
            f4 o = (f4){acc[nt][4 * g] + bv[nt], acc[nt][4 * g + 1] + bv[nt], acc[nt][4 * g + 2] + bv[nt],
                        acc[nt][4 * g + 3] + bv[nt]};
            f4* dst = reinterpret_cast<f4*>(y_nchw + (int64_t)n * geo.nchw_bstride + (int64_t)ch * HoWo + p);
            if (geo.yacc) o += *dst;
            *dst = o;
          }
        }
      } else {
        TileStat ts;
        ts.reset();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mbase + (r & 3) + 8 * (r >> 2);
          if (m < M) {
            float* dst = y.data + (int64_t)m * y.cstride + y.coff + ch;
            float v = acc[nt][r] + bv[nt];
            if (geo.yacc) v += *dst;
            *dst = v;
            ts.add(v);
          }
        }
        ts.flush(ssum[nt], ssq[nt]);
      }
    }
    __syncthreads();
  }

  if (stats) {
    double* redd = reinterpret_cast<double*>(red);       // [4][32*NT][2] doubles: 8 KB at NT = 4, inside the A tile
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const double s = ssum[nt] + __shfl_xor(ssum[nt], 32, 64);
      const double q = ssq[nt] + __shfl_xor(ssq[nt], 32, 64);
      if (lh == 0) {
        redd[(wave * 32 * NT + nt * 32 + l31) * 2 + 0] = s;
        redd[(wave * 32 * NT + nt * 32 + l31) * 2 + 1] = q;
      }
    }
    __syncthreads();
    if (tid < 32 * NT && tid < cout) {
      double s = 0, q = 0;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) {
        s += redd[(wv * 32 * NT + tid) * 2 + 0];
        q += redd[(wv * 32 * NT + tid) * 2 + 1];
      }
      double* st = stats + (size_t)(blockIdx.x % LHN_STAT_REPLICAS) * 2 * geo.statC;
      atomicAdd(st + tid, s);
      atomicAdd(st + geo.statC + tid, q);
    }
    if (fin.counter && lhn_last_block(fin.counter)) lhn_bn_finalize_block_par(fin, stats, redd);
  }
}

template <int CIN, int NT, int NS = 1>
static int launch_pw_fwd(const lhn_view* x, const float* w, const float* bias, const lhn_view* y, double* stats,
                         int stride, float* y_nchw, int cout, const lhn_bnfin* fin, const PwGeom& geo, hipStream_t s,
                         const PwExtra* exp = nullptr) {
  PwExtra ex;
  if (exp) ex = *exp; else { ex.n = 0; ex.sum_out = nullptr; }
  lhn_bnfin f;
  if (fin && stats) f = *fin; else f.counter = nullptr;
  const int M = y->N * y->H * y->W;
  const int ntiles = (M + 127) / 128;
  const size_t lds = (size_t)((32 * NT + 128) * (CIN + 4)) * sizeof(float);
  static LhnKernelCfg cfg;
  int per_cu = 1;
  if (!lhn_kernel_cfg(cfg, &k_pw_fwd<CIN, NT, NS>, lds, 4, &per_cu)) {
    lhn_set_error("lhn_conv_pw_fwd: cannot reserve %zu B of LDS", lds);
    return 2;
  }
  int grid = lhn_num_cus() * per_cu;
  if (grid > ntiles) grid = ntiles;
  hipLaunchKernelGGL((k_pw_fwd<CIN, NT, NS>), dim3(grid), dim3(256), lds, s, *x, w, bias, *y, stats, stride, y_nchw, cout, M,
                     ntiles, f, geo, ex);
  return 0;
}

// =====================================================================================================
// Forward, weights in REGISTERS (the fast path: stride 1, NHWC output, one input slice, Cout <= 128).
// A wave owns ONE 32-feature tile of W for the whole launch: its MFMA B fragments (CIN/2 VGPRs: lane = feature, lane half
// = k parity group) are loaded once from global memory.  LDS then only holds the pixel tile (double buffered, BM = 32..128
// pixels), 17..37 KB per block instead of 70..135 KB, so 2-3 blocks share a CU and one block's loads / LDS commit / stores
// run under another block's MFMA phase (the LDS-resident-W kernel above sits at ONE block per CU for 128 -> 128 and adds its
// phases up: 126 us against a 55 us MFMA bound).  Waves: NCOT feature tiles x (4 / NCOT) pixel sub-tiles of 32.
// Statistics need no cross-wave reduction: each wave owns its features.
template <int CIN, int NCOT, int NS>
__global__ void __launch_bounds__(256, (NS > 1 || CIN >= 128) ? 2 : 3)
k_pw_fwd_wr(lhn_view x, const float* __restrict__ w, const float* __restrict__ bias, lhn_view y, double* __restrict__ stats,
            int cout, int M, int ntiles, PwExtra ex, int wstride, int yacc, int statC, int wt, lhn_bnfin fin) {
  constexpr int LDA = CIN + 4, PXW = 4 / NCOT, BM = 32 * PXW;
  constexpr int C4 = CIN / 4, RP = 256 / C4, PF = BM / RP;      // float4 loads per thread, tile and source
  static_assert(PF >= 1, "tile too small for the loader");
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [2][BM][LDA]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lh = lane >> 5;
  const int cot = wave % NCOT, pxs = wave / NCOT;
  const int co = cot * 32 + l31;
  // ---- B fragments: wreg[kc*4 + j] = W[co][8*kc + 4*lh + j]  (same K permutation as the A reads below)
  float wreg[CIN / 2];
#pragma unroll
  for (int kc = 0; kc < CIN / 8; ++kc) {
    f4 v = (f4){0.f, 0.f, 0.f, 0.f};
    if (co < cout) {
      if (!wt) v = *reinterpret_cast<const f4*>(w + (int64_t)co * wstride + kc * 8 + 4 * lh);      // wstride: row of the WHOLE weight
      else {      // transposed use (data gradient: out[ci] = sum_co dy[co] * W[co][ci]): feature `co` of this launch is a COLUMN of W
        const float* wc = w + (int64_t)(kc * 8 + 4 * lh) * wstride + co;
        v = (f4){wc[0], wc[wstride], wc[2 * (int64_t)wstride], wc[3 * (int64_t)wstride]};
      }
    }
    wreg[kc * 4 + 0] = v.x; wreg[kc * 4 + 1] = v.y; wreg[kc * 4 + 2] = v.z; wreg[kc * 4 + 3] = v.w;
  }
  const float bv = (bias && co < cout) ? bias[co] : 0.f;
  // ---- loader geometry
  const int c4 = tid % C4, row0 = tid / C4;
  const int cabs = x.coff + 4 * c4;
  const int HoWo = y.H * y.W;
  int ecabs[NS > 1 ? NS - 1 : 1];
  if (NS > 1) {
#pragma unroll
    for (int e = 0; e < NS - 1; ++e)
      if (e < ex.n) ecabs[e] = ex.v[e].coff + 4 * c4;
  }
  f4 pre[PF], epre[NS > 1 ? NS - 1 : 1][PF];
  Xf4 xf, exf[NS > 1 ? NS - 1 : 1];          // filled after the first tile's loads have been issued (see below)
  // gates: one sample per tile whenever H*W is a multiple of the tile (every map of the models here but 8x8 at BM = 128), so the
  // gate of a tile is ONE float4 per thread, fetched with the tile's pixels a tile ahead (the general commit below loads it per
  // row and waits for it: a vmcnt(0) in the middle of the loop that also drained the prefetch)
  const bool uni = HoWo % BM == 0;
  const f4 one4 = (f4){1.f, 1.f, 1.f, 1.f};
  f4 gpre = one4, egpre[NS > 1 ? NS - 1 : 1];
  auto issue = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const int m = min(tile * BM + row0 + p * RP, M - 1);      // clamped: rows >= M are zeroed at commit
      pre[p] = *reinterpret_cast<const f4*>(x.data + (int64_t)m * x.cstride + cabs);
      if (NS > 1) {
#pragma unroll
        for (int e = 0; e < NS - 1; ++e)
          if (e < ex.n) epre[e][p] = *reinterpret_cast<const f4*>(ex.v[e].data + (int64_t)m * ex.v[e].cstride + ecabs[e]);
      }
    }
    if (uni) {
      const int n = min(tile * BM, M - 1) / HoWo;
      gpre = x.gate ? *reinterpret_cast<const f4*>(x.gate + (int64_t)n * x.cstride + cabs) : one4;
      if (NS > 1) {
#pragma unroll
        for (int e = 0; e < NS - 1; ++e)
          if (e < ex.n) egpre[e] = ex.v[e].gate ? *reinterpret_cast<const f4*>(ex.v[e].gate + (int64_t)n * ex.v[e].cstride + ecabs[e]) : one4;
      }
    }
  };
  auto commit = [&](int tile, float* As) __attribute__((always_inline)) {
    // one straight-line form: rows >= M were loaded from a clamped (valid) address and are zeroed by a select, not a branch
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const int row = row0 + p * RP, m = tile * BM + row;
      const bool ok = m < M;
      f4 g = gpre;
      if (!uni) {      // (a tile that spans samples: the gate per row)
        const int n = min(m, M - 1) / HoWo;
        g = x.gate ? *reinterpret_cast<const f4*>(x.gate + (int64_t)n * x.cstride + cabs) : one4;
      }
      f4 v = lhn_apply_xf(pre[p], xf) * g;
      if (NS > 1) {
        v *= ex.coef[0];
#pragma unroll
        for (int e = 0; e < NS - 1; ++e)
          if (e < ex.n) {
            f4 ge = egpre[e];
            if (!uni) {
              const int n = min(m, M - 1) / HoWo;
              ge = ex.v[e].gate ? *reinterpret_cast<const f4*>(ex.v[e].gate + (int64_t)n * ex.v[e].cstride + ecabs[e]) : one4;
            }
            v += lhn_apply_xf(epre[e][p], exf[e]) * ge * ex.coef[e + 1];
          }
        if (ex.sum_out && ok) *reinterpret_cast<f4*>(ex.sum_out + (int64_t)m * ex.so_cstride + ex.so_coff + 4 * c4) = v;
      }
      *reinterpret_cast<f4*>(As + row * LDA + 4 * c4) = ok ? v : (f4){0.f, 0.f, 0.f, 0.f};
    }
  };

  double ssum = 0.0, ssq = 0.0;       // fp32 partials per tile, promoted per tile (see k_pw_fwd)
  int tile = blockIdx.x, buf = 0;
  if (tile < ntiles) issue(tile);     // the first tile's loads are in flight while pending BatchNorms are finalized
  xf = lhn_load_xf(x, cabs);
  if (NS > 1) {
#pragma unroll
    for (int e = 0; e < NS - 1; ++e)
      if (e < ex.n) {
        exf[e] = lhn_load_xf(ex.v[e], ecabs[e]);
      }
  }
  if (tile < ntiles) {
    commit(tile, smem);
    if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x);
  }
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
    const float* As = smem + buf * BM * LDA;
    f16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* arow = As + (pxs * 32 + l31) * LDA + 4 * lh;
#pragma unroll
    for (int kc = 0; kc < CIN / 8; ++kc) {
      const f4 a = *reinterpret_cast<const f4*>(arow + kc * 8);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wreg[kc * 4 + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wreg[kc * 4 + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wreg[kc * 4 + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wreg[kc * 4 + 3], acc, 0, 0, 0);
    }
    // next tile: registers -> the other LDS buffer (nobody reads it during this iteration), then the loads of the tile after
    const int next = tile + gridDim.x;
    if (next < ntiles) {
      commit(next, smem + (buf ^ 1) * BM * LDA);
      if (next + (int)gridDim.x < ntiles) issue(next + gridDim.x);
    }
    // epilogue: C/D layout col = lane&31 (feature), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel)
    if (co < cout) {
      const int mbase = tile * BM + pxs * 32 + 4 * lh;
      float* yo = y.data + y.coff + co;
      TileStat ts;
      ts.reset();
      if (CIN < 256 && tile * BM + BM <= M && !yacc) {      // (K = 256 holds 128 registers of W: the batched stores below would spill)
        // whole tile inside the tensor (every tile but possibly the last): 16 plain stores off one base pointer.  The general
        // form below costs ~20 instructions and two branches per stored element (64-bit address product, row predicate, the
        // accumulate switch, the first-value select of TileStat) -- more issue slots than the tile's 16 MFMAs
        float* yp = yo + (int64_t)mbase * y.cstride;
        const int cs = y.cstride;
        const float k0 = acc[0] + bv;
        float sd = 0.f, qd = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[r] + bv;
          yp[((r & 3) + 8 * (r >> 2)) * cs] = v;
          const float d = v - k0;
          sd += d;
          qd += d * d;
        }
        ts.k = k0; ts.s = sd; ts.q = qd; ts.n = 16;
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mbase + (r & 3) + 8 * (r >> 2);
          if (m < M) {
            float v = acc[r] + bv;
            if (yacc) v += yo[(int64_t)m * y.cstride];       // second K slice of a wide input (LHN_PW_K256=0), or the data gradient of a
                                                             // 256-feature convolution accumulating into dx (lhn_pw_dgrad_wr): statistics see the sum
            yo[(int64_t)m * y.cstride] = v;
            ts.add(v);
          }
        }
      }
      ts.flush(ssum, ssq);
    }
    __syncthreads();
  }
  if (stats && co < cout) {
    const double s = ssum + __shfl_xor(ssum, 32, 64), q = ssq + __shfl_xor(ssq, 32, 64);
    if (lh == 0) {
      double* st = stats + (size_t)(blockIdx.x % LHN_STAT_REPLICAS) * 2 * statC;      // statC: channels of the whole BatchNorm
      atomicAdd(st + co, s);
      atomicAdd(st + statC + co, q);
    }
  }
  // fused BatchNorm finalize (single-launch convolutions only, see pw_fwd_slice): the pixel tiles in LDS are dead by now
  if (stats && fin.counter && lhn_last_block(fin.counter)) lhn_bn_finalize_block_par(fin, stats, reinterpret_cast<double*>(smem));
}

template <int CIN, int NCOT, int NS>
static int launch_pw_fwd_wr(const lhn_view* x, const float* w, const float* bias, const lhn_view* y, double* stats, int cout,
                            hipStream_t s, const PwExtra* exp, const PwGeom& geo, int wt = 0, const lhn_bnfin* finp = nullptr) {
  PwExtra ex;
  if (exp) ex = *exp; else { ex.n = 0; ex.sum_out = nullptr; }
  constexpr int BM = 32 * (4 / NCOT);
  const int M = y->N * y->H * y->W, ntiles = (M + BM - 1) / BM;
  const size_t lds = (size_t)2 * BM * (CIN + 4) * sizeof(float);
  static LhnKernelCfg cfg;
  int per_cu = 1;
  if (!lhn_kernel_cfg(cfg, &k_pw_fwd_wr<CIN, NCOT, NS>, lds, 4, &per_cu)) {
    lhn_set_error("lhn_conv_pw_fwd: cannot reserve %zu B of LDS", lds);
    return 2;
  }
  int grid = lhn_num_cus() * per_cu;      // (2 / 4 / 8 / 16 blocks per CU measured in round 3: forward 2.71 / 2.69 / 2.74 / 2.83 ms)
  if (grid > ntiles) grid = ntiles;
  lhn_bnfin fin;
  if (finp) fin = *finp; else fin.counter = nullptr;
  hipLaunchKernelGGL((k_pw_fwd_wr<CIN, NCOT, NS>), dim3(grid), dim3(256), lds, s, *x, w, bias, *y, stats, cout, M, ntiles, ex, geo.wstride,
                     geo.yacc, geo.statC, wt, fin);
  return 0;
}

// returns -1 when the register-resident-weights kernel has no instance for the shape
static int pw_fwd_wr(const lhn_view* x, const float* w, const float* bias, const lhn_view* y, double* stats, int cout, hipStream_t s,
                     const PwExtra* ex, const PwGeom& geo, int wt = 0, const lhn_bnfin* fin = nullptr) {
  static int off = -1;
  if (off < 0) {
    const char* e = getenv("LHN_PW_LDSW");      // 1 = always take the LDS-resident-W kernel (A/B comparisons)
    off = (e && e[0] == '1') ? 1 : 0;
  }
  if (off) return -1;
  const int ci = x->C, ncot = cout <= 32 ? 1 : cout <= 64 ? 2 : 4;
  const bool ms = ex && ex->n > 0;
  // K = 256 (hourglassnet.py: every 1x1 of the C = 256 residuals) in ONE pass: 128 VGPRs of W per lane, no second K slice that
  // re-reads and re-writes y
  if (ci == 256 && !ms) {
    if (ncot == 4) return launch_pw_fwd_wr<256, 4, 1>(x, w, bias, y, stats, cout, s, ex, geo, wt, fin);
    return -1;      // (64 features per block would need 64-pixel tiles: 256 VGPRs and spills)
  }
  if (ms) {
    if (ci == 128 && ncot == 4) return launch_pw_fwd_wr<128, 4, 3>(x, w, bias, y, stats, cout, s, ex, geo, 0, fin);
    if (ci == 64 && ncot == 2) return launch_pw_fwd_wr<64, 2, 3>(x, w, bias, y, stats, cout, s, ex, geo, 0, fin);
    return -1;
  }
#define WR(CI, NC) if (ci == CI && ncot == NC) return launch_pw_fwd_wr<CI, NC, 1>(x, w, bias, y, stats, cout, s, ex, geo, wt, fin);
  WR(128, 4) WR(128, 2) WR(64, 4) WR(64, 2) WR(64, 1) WR(32, 4) WR(32, 2) WR(32, 1)
#undef WR
  return -1;
}

// Data gradient of a 1x1 on the same kernel: dx[.., ci0 + n] (+)= sum_k dy[.., k] * W[co0 + k][ci0 + n].  dyv: the plain dy
// (k_dy_inplace ran) as a view of C = 32/64/128 output features, dxv: <= 128 input channels of the gradient buffer,
// w = &W[co0][ci0], wstride = Cin of the whole weight.  Returns -1 when there is no instance for the shape.
int lhn_pw_dgrad_wr(const lhn_view* dyv, const float* w, const lhn_view* dxv, int wstride, int accumulate, hipStream_t s) {
  static int off = -1;
  if (off < 0) {
    const char* e = getenv("LHN_PW_LDSW");
    off = (e && e[0] == '1') ? 1 : 0;
  }
  if (off) return -1;
  PwGeom g;
  g.wstride = wstride;
  g.kvalid = dyv->C;
  g.wrows = dxv->C;
  g.yacc = accumulate;
  g.statC = dxv->C;
  g.nchw_bstride = 0;
  return pw_fwd_wr(dyv, w, nullptr, dxv, nullptr, dxv->C, s, nullptr, g, 1);
}

// smallest tile width (16/32/64/128) that holds `c` input channels; 0 = none
static inline int pw_cin_tile(int c) { return c <= 16 ? 16 : c <= 32 ? 32 : c <= 64 ? 64 : c <= 128 ? 128 : 0; }

// one (input slice, output slice) launch
static int pw_fwd_slice(const lhn_view* x, const float* w, const float* bias, const lhn_view* y, double* stats, int stride,
                        float* y_nchw, int cout, const lhn_bnfin* fin, const PwGeom& geo, hipStream_t s, const PwExtra* ex = nullptr) {
  const int ci = x->C == 256 ? 256 : pw_cin_tile(x->C), nt = (cout + 31) / 32 == 3 ? 4 : (cout + 31) / 32;
  int rc = -1;
  // fast path: whole-K slice with full-width rows, plain NHWC store (a fused finalize only when this launch is the whole conv)
  // (slices of a C = 256 convolution qualify too: the kernel takes the whole weight's row stride, accumulates later K slices
  // into y and addresses the statistics of the whole BatchNorm)
  if (stride == 1 && !y_nchw && x->C == ci && geo.kvalid == ci && geo.wstride % 4 == 0 && geo.wrows == cout &&
      !(geo.yacc && ex && ex->n > 0) && !(fin && stats && (geo.yacc || geo.statC != cout)) && (reinterpret_cast<uintptr_t>(w) & 15) == 0) {
    rc = pw_fwd_wr(x, w, bias, y, stats, cout, s, ex, geo, 0, stats ? fin : nullptr);
    if (rc != -1) return rc;
  }
  if (ex && ex->n > 0) {      // summed-on-load sources: the square 1x1 of MSRB (litehourglass.py:30,49), C = 64 / 128
    if (ci == 128 && nt == 4) rc = launch_pw_fwd<128, 4, 3>(x, w, bias, y, stats, stride, y_nchw, cout, fin, geo, s, ex);
    else if (ci == 64 && nt == 2) rc = launch_pw_fwd<64, 2, 3>(x, w, bias, y, stats, stride, y_nchw, cout, fin, geo, s, ex);
    return rc;
  }
#define PW_CASE(CI, NTV) \
  if (ci == CI && nt == NTV) rc = launch_pw_fwd<CI, NTV>(x, w, bias, y, stats, stride, y_nchw, cout, fin, geo, s, ex);
  PW_CASE(32, 1) PW_CASE(32, 2) PW_CASE(32, 4) PW_CASE(64, 1) PW_CASE(64, 2) PW_CASE(64, 4) PW_CASE(128, 1)
  PW_CASE(128, 2) PW_CASE(128, 4) PW_CASE(16, 1) PW_CASE(16, 2) PW_CASE(16, 4)
#undef PW_CASE
  return rc;
}

extern "C" int lhn_conv_pw_fwd2(const lhn_view* x, const float* w, const float* bias, const lhn_view* y, double* stats,
                                int stride, float* y_nchw, const lhn_bnfin* fin, const lhn_pw_opts* opts, void* stream) {
  LHN_CHECK_ARG(lhn_view_ok(x) && w && y, "lhn_conv_pw_fwd: bad input view / null pointer");
  LHN_CHECK_ARG(stride == 1 || stride == 2, "lhn_conv_pw_fwd: stride %d", stride);
  LHN_CHECK_ARG(y->N == x->N && y->H == (x->H + stride - 1) / stride && y->W == (x->W + stride - 1) / stride,
                "lhn_conv_pw_fwd: output geometry %dx%dx%d does not match input %dx%dx%d / stride %d", y->N, y->H, y->W,
                x->N, x->H, x->W, stride);
  const int Cout = y->C, Cin = x->C, HoWo = y->H * y->W;
  if (y_nchw) {
    LHN_CHECK_ARG(HoWo % 4 == 0 && Cout > 0, "lhn_conv_pw_fwd: NCHW output needs H*W %% 4 == 0");
    LHN_CHECK_ARG(!stats, "lhn_conv_pw_fwd: no statistics on the NCHW head");
  } else {
    LHN_CHECK_ARG(lhn_view_ok(y), "lhn_conv_pw_fwd: bad output view");
  }
  LHN_CHECK_ARG((int64_t)y->N * y->H * y->W < (1ll << 31) - 256, "lhn_conv_pw_fwd: too many pixels");
  // weight tensor: [w_rows][w_cols], w_cols real input channels (<= Cin, the view may be padded to a multiple of 4),
  // w_rows real output features (<= Cout)
  const int wcols = (opts && opts->w_cols > 0) ? opts->w_cols : Cin, wrows = (opts && opts->w_rows > 0) ? opts->w_rows : Cout;
  LHN_CHECK_ARG(wcols <= Cin && wcols > Cin - 4 && wrows <= Cout, "lhn_conv_pw_fwd: weight [%d][%d] does not fit views %d -> %d",
                wrows, wcols, Cin, Cout);
  const int64_t bstride = (opts && opts->nchw_batch_stride > 0) ? opts->nchw_batch_stride : (int64_t)Cout * HoWo;
  hipStream_t s = (hipStream_t)stream;
  const bool single = Cin <= 128 && Cout <= 128;
  PwExtra ex;
  ex.n = 0;
  ex.sum_out = nullptr;
  ex.so_cstride = ex.so_coff = 0;
  LHN_CHECK_ARG(lhn_no_pend(x), "lhn_conv_pw_fwd: lhn_view.pend is reserved (NULL)");
  if (opts && opts->n_extra > 0) {
    LHN_CHECK_ARG(opts->n_extra <= 2 && opts->extra && single && stride == 1, "lhn_conv_pw_fwd: extra sources need stride 1 and <= 128 channels");
    ex.n = opts->n_extra;
    for (int e = 0; e < ex.n; ++e) {
      const lhn_view* v = &opts->extra[e];
      LHN_CHECK_ARG(lhn_view_ok(v) && v->C == Cin && v->N == x->N && v->H == x->H && v->W == x->W && lhn_no_pend(v), "lhn_conv_pw_fwd: extra source %d geometry", e);
      ex.v[e] = *v;
    }
    for (int e = 0; e < 3; ++e) ex.coef[e] = opts->coef[e];
    if (opts->sum_out) {
      const lhn_view* so = opts->sum_out;
      LHN_CHECK_ARG(so->data && so->C == Cin && so->N == x->N && so->H == x->H && so->W == x->W && so->cstride % 4 == 0 && so->coff % 4 == 0 &&
                        so->coff + so->C <= so->cstride && Cout <= 128 && Cin % 4 == 0,
                    "lhn_conv_pw_fwd: sum_out geometry (same pixels and channels as x, one output slice)");
      ex.sum_out = so->data;
      ex.so_cstride = so->cstride;
      ex.so_coff = so->coff;
    }
  } else {
    LHN_CHECK_ARG(!(opts && opts->sum_out), "lhn_conv_pw_fwd: sum_out without extra sources");
  }
  // Cin = 256 with full-width rows: the register-W kernel takes the whole K at once (LHN_PW_K256=0: two K slices, the second
  // one accumulating into y)
  static int k256 = -1;
  if (k256 < 0) {
    const char* e = getenv("LHN_PW_K256");
    k256 = (e && e[0] == '0') ? 0 : 1;
  }
  const int kstep = (k256 && Cin == 256 && wcols == 256 && stride == 1 && !y_nchw && ex.n == 0 && Cout % 128 == 0 && wrows == Cout &&
                     (reinterpret_cast<uintptr_t>(w) & 15) == 0 && !(fin && stats && single)) ? 256 : 128;
  for (int co0 = 0; co0 < Cout; co0 += 128) {
    const int cc = Cout - co0 < 128 ? Cout - co0 : 128;
    for (int k0 = 0; k0 < Cin; k0 += kstep) {
      const int kc = Cin - k0 < kstep ? Cin - k0 : kstep;
      const bool last = k0 + kc >= Cin;
      lhn_view xv = *x, yv = *y;
      xv.coff += k0;
      xv.C = kc;
      if (!y_nchw) {
        yv.coff += co0;
        yv.C = cc;
      }
      PwGeom g;
      g.wstride = wcols;
      g.kvalid = wcols - k0 < kc ? wcols - k0 : kc;
      g.wrows = wrows - co0 < cc ? (wrows - co0 < 0 ? 0 : wrows - co0) : cc;
      g.yacc = k0 > 0;
      g.statC = Cout;
      g.nchw_bstride = bstride;
      const int rc = pw_fwd_slice(&xv, w + (int64_t)co0 * wcols + k0, (last && bias) ? bias + co0 : nullptr, &yv,
                                  (last && stats) ? stats + co0 : nullptr, stride, y_nchw ? y_nchw + (int64_t)co0 * HoWo : nullptr,
                                  cc, single ? fin : nullptr, g, s, &ex);
      LHN_CHECK_ARG(rc != -1, "lhn_conv_pw_fwd: unsupported channels Cin=%d Cout=%d (%d extra sources)", Cin, Cout, ex.n);
      if (rc) return rc;
    }
  }
  LHN_CHECK_LAUNCH("lhn_conv_pw_fwd");
  if (!single && fin && stats) {   // fused finalize was requested: the sliced form runs it as its own launch
    return lhn_bn_finalize(stats, fin->gamma, fin->beta, fin->running_mean, fin->running_var, fin->num_batches_tracked, fin->table,
                           fin->cstride, fin->coff, fin->C, fin->save_mean_invstd, fin->count, fin->eps, fin->momentum, fin->slope,
                           1, fin->conv_bias, stream);
  }
  return 0;
}

extern "C" int lhn_conv_pw_fwd(const lhn_view* x, const float* w, const float* bias, const lhn_view* y, double* stats,
                               int stride, float* y_nchw, const lhn_bnfin* fin, void* stream) {
  return lhn_conv_pw_fwd2(x, w, bias, y, stats, stride, y_nchw, fin, nullptr, stream);
}

// =====================================================================================================
// Backward: fused dgrad + wgrad.  Per 64-pixel tile the block stages
//     dYs[m][co] = dy  (formed on the fly from dz, the saved raw output y and the BN-backward coefficients)
//     Xs [m][ci] = consumed input value (raw input + the producer's pending transform / gate)
// and issues   dX[m][ci] = sum_co dYs[m][co] * W[co][ci]   (K = Cout)   -> global, store or accumulate
//              dW[co][ci] += sum_m dYs[m][co] * Xs[m][ci]  (K = 64)     -> registers across tiles, one
//                                                                           fp32 atomic add per block at the end
// NCHW = the head's gradient arrives as a plain NCHW tensor (dy_nchw); a template flag so that the NHWC instances do not
// carry its prefetch registers (they cost k_pw_bwd<64,2> 44 -> 56 us when the switch was a runtime one)
#ifndef LHN_PWB_BNS_OCC
#define LHN_PWB_BNS_OCC 3      // (2 = no spills: B step 8.45 vs 8.41 ms -- the 9 spilled registers cost less than the third workgroup gives)
#endif
template <int CIN, int NTO, bool NCHW, bool BNS = false>
__global__ void __launch_bounds__(256, (CIN == 32 && NTO == 1 && !NCHW) ? 4 : (CIN <= 64 && NTO <= 2 && !NCHW) ? (BNS ? LHN_PWB_BNS_OCC : 3) : 1) k_pw_bwd(lhn_view x, const float* __restrict__ w, lhn_view y, lhn_gradview gy,
                                                float* __restrict__ dx, int dx_acc, float* __restrict__ dw,
                                                float* __restrict__ dbias, int stride, const float* __restrict__ dy_nchw,
                                                int cout, int M, int ntiles, int nrep, int64_t rep_stride, PwGeom geo, lhn_bnsum bs) {
  constexpr int NTI = CIN / 32;
  constexpr int COP = 32 * NTO;
  constexpr int LDW = CIN + 4, LDY = COP + 4, LDX = CIN + 4;
  // MFMA work split: dX has 2*NTI tiles of 16*NTO MFMAs, dW has T = NTO*NTI tiles of 32 MFMAs (equal totals).  With
  // CIN = 32 (NTI == 1) dX only occupies waves 0,1, so dW goes to waves 2,3; when there are fewer dW tiles than dW
  // waves the 64-pixel K range of a tile is split across waves (partial sums meet in the final atomic flush).
  constexpr int T = NTO * NTI;
  constexpr bool DW_HI = (NTI == 1);
  constexpr int DWW = DW_HI ? 2 : 4;                   // waves that compute dW
  constexpr int KS = T < DWW ? DWW / T : 1;            // K parts per dW tile
  constexpr int NDW = (T * KS + DWW - 1) / DWW;        // dW accumulators per wave
  constexpr int NDX = (NTI + 1) / 2;                   // dX ci-tiles per wave
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                 // [COP][LDW]
  float* dYs = Ws + COP * LDW;      // [64][LDY]
  float* Xs = dYs + 64 * LDY;       // [64][LDX]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lh = lane >> 5;
  const int HoWo = y.H * y.W;

  pw_stage_w<CIN>(Ws, LDW, w, COP, geo);

  // loader geometry: X tile 64 x CIN/4 float4, dY tile 64 x COP/4 float4
  constexpr int XC4 = CIN / 4, XRP = 256 / XC4, XPF = 64 / XRP;
  constexpr int YC4 = COP / 4, YRP = 256 / YC4, YPF = 64 / YRP;
  const int xc4 = tid % XC4, xr0 = tid / XC4;
  const bool xok = 4 * xc4 < x.C;      // channel groups beyond the input slice are zero columns
  const int xabs = x.coff + (xok ? 4 * xc4 : 0);
  const int yc4 = tid % YC4, yr0 = tid / YC4, yabs = y.coff + 4 * yc4;
  const Xf4 xxf = lhn_load_xf(x, xabs);
  const bool ych_ok = 4 * yc4 < cout;  // cout is a multiple of 4 on the NHWC path
  Xf4 yxf;
  Gr4 ygr;
  if (!NCHW && ych_ok) {
    yxf = lhn_load_xf(y, yabs);
    ygr = lhn_load_coef(gy, y.cstride, yabs);
  }
  f4 bsum = (f4){0.f, 0.f, 0.f, 0.f};

  f16v accw[NDW];
#pragma unroll
  for (int t = 0; t < NDW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[t][r] = 0.f;

  auto in_pix = [&](int m) __attribute__((always_inline)) -> int64_t {
    if (stride == 1) return m;
    const int n = m / HoWo, r = m - n * HoWo;
    const int ho = r / y.W, wo = r - ho * y.W;
    return ((int64_t)n * x.H + ho * stride) * x.W + wo * stride;
  };

  // global loads of a tile go to registers (issue) one iteration ahead of their transform + LDS store (commit), so the
  // next tile's HBM latency overlaps this tile's MFMA work
  f4 xraw[XPF], yraw[YPF], ydz[YPF];
  constexpr int NCHW_PF = NCHW ? 64 * COP / 256 : 1;   // NCHW head: dY elements per thread and tile, prefetched like the rest
  float ynch[NCHW_PF];
  auto issue = [&](int tile) __attribute__((always_inline)) {
    if constexpr (NCHW) {
#pragma unroll
      for (int k = 0; k < NCHW_PF; ++k) {
        const int i = tid + 256 * k, row = i & 63, co = i >> 6, m = min(tile * 64 + row, M - 1);
        const int n = m / HoWo, p = m - n * HoWo;
        ynch[k] = co < cout ? dy_nchw[(int64_t)n * geo.nchw_bstride + (int64_t)co * HoWo + p] : 0.f;
      }
    }
#pragma unroll
    for (int p = 0; p < XPF; ++p) {
      const int m = min(tile * 64 + xr0 + p * XRP, M - 1);      // clamped (branch-free); commit() zeroes rows >= M
      xraw[p] = *reinterpret_cast<const f4*>(x.data + in_pix(m) * x.cstride + xabs);
    }
    if (!NCHW && ych_ok) {
#pragma unroll
      for (int p = 0; p < YPF; ++p) {
        const int m = min(tile * 64 + yr0 + p * YRP, M - 1);
        yraw[p] = *reinterpret_cast<const f4*>(y.data + (int64_t)m * y.cstride + yabs);
        ydz[p] = *reinterpret_cast<const f4*>(gy.dz + (int64_t)m * y.cstride + yabs);
      }
    }
  };
  auto commit = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < XPF; ++p) {
      const int row = xr0 + p * XRP, m = tile * 64 + row;
      f4 v = (f4){0.f, 0.f, 0.f, 0.f};
      if (m < M && xok) {
        v = lhn_apply_xf(xraw[p], xxf);
        if (x.gate) v *= *reinterpret_cast<const f4*>(x.gate + (int64_t)(m / HoWo) * x.cstride + xabs);
      }
      *reinterpret_cast<f4*>(Xs + row * LDX + 4 * xc4) = v;
    }
    if constexpr (NCHW) {
#pragma unroll
      for (int k = 0; k < NCHW_PF; ++k) {
        const int i = tid + 256 * k, row = i & 63, co = i >> 6;
        dYs[row * LDY + co] = (tile * 64 + row < M) ? ynch[k] : 0.f;
      }
    } else {
#pragma unroll
      for (int p = 0; p < YPF; ++p) {
        const int row = yr0 + p * YRP, m = tile * 64 + row;
        f4 v = (f4){0.f, 0.f, 0.f, 0.f};
        if (m < M && ych_ok) {
          const int n = m / HoWo, r = m - n * HoWo;
          const int h = r / y.W, ww = r - h * y.W;
          const f4 du = lhn_grad_du(y, gy, yxf, yraw[p], ydz[p], n, h, ww, yabs);
          v = ygr.A * du + ygr.B * yraw[p] + ygr.Cc;
          bsum += v;
        }
        *reinterpret_cast<f4*>(dYs + row * LDY + 4 * yc4) = v;
      }
    }
  };

  // bs.sums (lhn_bnsum): this launch's dX is (part of) the gradient of the value of x, the output of a convolution + BatchNorm
  // -- the lane that holds dX[m][ci] re-reads the raw x[m][ci] (the tile was just staged: L2) and adds du = dX * act'(u) and
  // du * xhat to its channel's sums; stride 1, host-checked.  Channel of lane and dX tile t: ci = 32 * ((wave >> 1) + 2 t) + l31.
  // (BNS is a template flag: as a runtime one its 14 registers spilled k_pw_bwd<64,2>)
  constexpr int NB = BNS ? NDX : 1;
  float b_sc[NB], b_sh[NB], b_sl[NB], b_mean[NB], b_inv[NB], b_s[NB], b_q[NB];
#pragma unroll
  for (int t = 0; t < NB; ++t) {
    b_s[t] = b_q[t] = 0.f;
    const int ci = 32 * ((wave >> 1) + 2 * t) + l31;
    const bool ok = BNS && bs.sums && ci < x.C;
    b_sc[t] = ok && x.table ? x.table[x.coff + ci] : 1.f;
    b_sh[t] = ok && x.table ? x.table[x.cstride + x.coff + ci] : 0.f;
    b_sl[t] = ok && x.table ? x.table[2 * x.cstride + x.coff + ci] : 1.f;
    b_mean[t] = ok ? bs.save[bs.coff + ci] : 0.f;
    b_inv[t] = ok ? bs.save[bs.C + bs.coff + ci] : 0.f;
  }
  int tile = blockIdx.x;
  if (tile < ntiles) issue(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    commit(tile);
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x);

    // ---- dW += dY^T X   (K = 64 pixels)
    const int dwave = DW_HI ? wave - 2 : wave;
    if constexpr (!DW_HI && KS == 1 && T % 4 == 0 && 4 % NTI == 0) {
      // every wave owns whole dW tiles (tile wave + 4 t: row wave / NTI + (4 / NTI) t, column wave % NTI): no predicate around
      // the MFMAs -- with `if (tl < T)` here the compiler wrapped each one in exec-mask branches and lgkmcnt(0) waits (k_conv_kxk.hip)
      const float* ap = dYs + lh * LDY + l31 + 32 * (wave / NTI);
      const float* bp = Xs + lh * LDX + l31 + 32 * (wave % NTI);
#pragma unroll 4
      for (int ks = 0; ks < 32; ++ks) {
        const float b = bp[(2 * ks) * LDX];
        float a[NDW];
#pragma unroll
        for (int t = 0; t < NDW; ++t) a[t] = ap[(2 * ks) * LDY + 32 * (4 / NTI) * t];
#pragma unroll
        for (int t = 0; t < NDW; ++t) accw[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b, accw[t], 0, 0, 0);
      }
    } else if (dwave >= 0) {
      const int kpart = KS > 1 ? dwave % KS : 0;
#pragma unroll 4
      for (int ks = kpart * (32 / KS); ks < (kpart + 1) * (32 / KS); ++ks) {
        const float* dyr = dYs + (2 * ks + lh) * LDY + l31;
        const float* xr = Xs + (2 * ks + lh) * LDX + l31;
#pragma unroll
        for (int t = 0; t < NDW; ++t) {
          const int tl = (dwave + DWW * t) / KS;
          if (tl < T) {
            const int it = tl / NTI, jt = tl % NTI;
            accw[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(dyr[32 * it], xr[32 * jt], accw[t], 0, 0, 0);
          }
        }
      }
    }
    // ---- dX = dY W   (K = Cout)
    if (dx) {
      const int mt = wave & 1;
      f16v accx[NDX];
#pragma unroll
      for (int t = 0; t < NDX; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) accx[t][r] = 0.f;
      const float* arow = dYs + (mt * 32 + l31) * LDY + 4 * lh;
#pragma unroll 2
      for (int kc = 0; kc < COP / 8; ++kc) {
        const f4 a = *reinterpret_cast<const f4*>(arow + kc * 8);
        const float* wr = Ws + (kc * 8 + 4 * lh) * LDW + l31;
#pragma unroll
        for (int t = 0; t < NDX; ++t) {
          const int jt = (wave >> 1) + 2 * t;
          if (NTI % 2 == 0 || jt < NTI) {        // (even NTI: always true, folded at compile time)
            accx[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wr[0 * LDW + 32 * jt], accx[t], 0, 0, 0);
            accx[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wr[1 * LDW + 32 * jt], accx[t], 0, 0, 0);
            accx[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wr[2 * LDW + 32 * jt], accx[t], 0, 0, 0);
            accx[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wr[3 * LDW + 32 * jt], accx[t], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int t = 0; t < NDX; ++t) {
        const int jt = (wave >> 1) + 2 * t;
        if (jt < NTI) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = tile * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m < M && 32 * jt + l31 < x.C) {
              float* o = dx + in_pix(m) * x.cstride + x.coff + 32 * jt + l31;
              if (BNS) {
                const float raw = x.data[(int64_t)m * x.cstride + x.coff + 32 * jt + l31];
                const float du = accx[t][r] * (raw * b_sc[t] + b_sh[t] > 0.f ? 1.f : b_sl[t]);
                b_s[t] += du;
                b_q[t] += du * ((raw - b_mean[t]) * b_inv[t]);
              }
              *o = dx_acc ? *o + accx[t][r] : accx[t][r];
            }
          }
        }
      }
    }
    __syncthreads();
  }

  if (BNS) {      // the two lane halves and the two waves (pixel halves) of a channel tile meet in LDS (Xs is free now)
    float* bred = Xs;                                   // [NTI][2 waves][2][32]
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      const int jt = (wave >> 1) + 2 * t;
      const float ss = b_s[t] + __shfl_xor(b_s[t], 32, 64), qq = b_q[t] + __shfl_xor(b_q[t], 32, 64);
      if (jt < NTI && lh == 0) {
        bred[((jt * 2 + (wave & 1)) * 2 + 0) * 32 + l31] = ss;
        bred[((jt * 2 + (wave & 1)) * 2 + 1) * 32 + l31] = qq;
      }
    }
    __syncthreads();
    if (tid < NTI * 32 && tid < x.C) {
      const int jt = tid >> 5, c = tid & 31;
      double* st = bs.sums + (size_t)(blockIdx.x % LHN_STAT_REPLICAS) * 2 * bs.C + bs.coff + tid;
      atomicAdd(st, (double)bred[((jt * 2 + 0) * 2 + 0) * 32 + c] + (double)bred[((jt * 2 + 1) * 2 + 0) * 32 + c]);
      atomicAdd(st + bs.C, (double)bred[((jt * 2 + 0) * 2 + 1) * 32 + c] + (double)bred[((jt * 2 + 1) * 2 + 1) * 32 + c]);
    }
    __syncthreads();
  }
  // ---- flush dW (C/D layout: row = co within tile, col = lane&31 = ci within tile)
  dw += (size_t)(blockIdx.x % nrep) * rep_stride;
#pragma unroll
  for (int t = 0; t < NDW; ++t) {
    const int dwave = DW_HI ? wave - 2 : wave;
    const int tl = dwave >= 0 ? (dwave + DWW * t) / KS : T;
    if (tl < T) {
      const int it = tl / NTI, jt = tl % NTI;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = 32 * it + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (co < geo.wrows && 32 * jt + l31 < geo.kvalid) atomicAdd(dw + (int64_t)co * geo.wstride + 32 * jt + l31, accw[t][r]);
      }
    }
  }
  if (dbias) {
    if (NCHW) {
      // head: few channels; recompute per-channel sums is cheap -- done by a separate tiny pass on the host side
    } else {
      // column sums of dy: the threads that share a channel group meet in LDS in a FIXED order (dYs is free by now), then
      // one add per channel into this block's gradient replica -- a single writer per address, like dW
      f4* bred = reinterpret_cast<f4*>(dYs);            // [YRP][YC4] float4 <= 64 * LDY floats
      bred[yr0 * YC4 + yc4] = ych_ok ? bsum : (f4){0.f, 0.f, 0.f, 0.f};
      __syncthreads();
      if (tid < YC4 && 4 * tid < cout) {
        f4 t = (f4){0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < YRP; ++j) t += bred[j * YC4 + tid];
        float* db = dbias + (size_t)(blockIdx.x % nrep) * rep_stride + 4 * tid;
        if (4 * tid + 0 < geo.wrows) atomicAdd(db + 0, t.x);
        if (4 * tid + 1 < geo.wrows) atomicAdd(db + 1, t.y);
        if (4 * tid + 2 < geo.wrows) atomicAdd(db + 2, t.z);
        if (4 * tid + 3 < geo.wrows) atomicAdd(db + 3, t.w);
      }
    }
  }
}

// bias gradient of the NCHW head: db[co] = sum_{n,p} dy[n,co,p]
__global__ void __launch_bounds__(256) k_bias_grad_nchw(const float* __restrict__ dy, float* __restrict__ db, int N, int C,
                                                        int HW, int64_t bstride) {
  const int co = blockIdx.x;
  double s = 0;
  for (int n = blockIdx.y; n < N; n += gridDim.y) {
    const float* p = dy + (int64_t)n * bstride + (int64_t)co * HW;
    float a = 0.f;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) a += p[i];
    s += a;
  }
  s = lhn_wave_sum_d(s);
  __shared__ double red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(db + co, (float)(red[0] + red[1] + red[2] + red[3]));
}

template <int CIN, int NTO, bool NCHW, bool BNS = false>
static int launch_pw_bwd_t(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, float* dx, int dx_acc,
                         float* dw, float* dbias, int stride, const float* dy_nchw, int cout, int nrep, int64_t rep_stride,
                         const PwGeom& geo, hipStream_t s, const lhn_bnsum& bs) {
  const int M = y->N * y->H * y->W;
  const int ntiles = (M + 63) / 64;
  constexpr int COP = 32 * NTO;
  const size_t lds = (size_t)(COP * (CIN + 4) + 64 * (COP + 4) + 64 * (CIN + 4)) * sizeof(float);
  static LhnKernelCfg cfg;
  int per_cu = 1;
  if (!lhn_kernel_cfg(cfg, &k_pw_bwd<CIN, NTO, NCHW, BNS>, lds, 4, &per_cu)) {
    lhn_set_error("lhn_conv_pw_bwd: cannot reserve %zu B of LDS", lds);
    return 2;
  }
  int grid = lhn_num_cus() * per_cu;
  if (grid > ntiles) grid = ntiles;
  // (one tile per workgroup on small maps is the fastest split: fewer, longer workgroups -- 2 / 4 / 8 tiles each -- took Lite-HRNet's
  // step from 36.9 to 40.1 / 47.4 / 55.3 ms; the weight-gradient atomics are not what these launches wait for)
  lhn_gradview g = *gy;
  hipLaunchKernelGGL((k_pw_bwd<CIN, NTO, NCHW, BNS>), dim3(grid), dim3(256), lds, s, *x, w, *y, g, dx, dx_acc, dw, dbias, stride, dy_nchw,
                     cout, M, ntiles, nrep, rep_stride, geo, bs);
  if (dbias && dy_nchw)
    hipLaunchKernelGGL(k_bias_grad_nchw, dim3(geo.wrows, lhn_deterministic_mode() ? 1 : (y->N < 16 ? y->N : 16)), dim3(256), 0, s, dy_nchw, dbias, y->N, cout,
                       y->H * y->W, geo.nchw_bstride);
  return 0;
}

template <int CIN, int NTO>
static int launch_pw_bwd(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, float* dx, int dx_acc,
                         float* dw, float* dbias, int stride, const float* dy_nchw, int cout, int nrep, int64_t rep_stride,
                         const PwGeom& geo, hipStream_t s, const lhn_bnsum& bs) {
  if (dy_nchw) return launch_pw_bwd_t<CIN, NTO, true>(x, w, y, gy, dx, dx_acc, dw, dbias, stride, dy_nchw, cout, nrep, rep_stride, geo, s, bs);
  if constexpr (CIN * NTO * 32 < 64 * 128) {           // reader-side BatchNorm sums: the shapes the fused kernel keeps for itself
    if (bs.sums && dx && stride == 1)
      return launch_pw_bwd_t<CIN, NTO, false, true>(x, w, y, gy, dx, dx_acc, dw, dbias, stride, dy_nchw, cout, nrep, rep_stride, geo, s, bs);
  }
  if (bs.sums) return -1;
  return launch_pw_bwd_t<CIN, NTO, false>(x, w, y, gy, dx, dx_acc, dw, dbias, stride, dy_nchw, cout, nrep, rep_stride, geo, s, bs);
}

int lhn_pw_bwd_split(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, float* dx, int dx_accumulate,
                     float* dw, float* dbias, int nrep, int64_t rep_stride, hipStream_t s);

extern "C" int lhn_conv_pw_bwd2(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, float* dx,
                                int dx_accumulate, float* dw, float* dbias, int stride, const float* dy_nchw, int nrep,
                                int64_t rep_stride, const lhn_pw_opts* opts, void* stream) {
  return lhn_conv_pw_bwd3(x, w, y, gy, dx, dx_accumulate, dw, dbias, stride, dy_nchw, nrep, rep_stride, opts, nullptr, stream);
}
extern "C" int lhn_conv_pw_bwd3(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, float* dx,
                                int dx_accumulate, float* dw, float* dbias, int stride, const float* dy_nchw, int nrep,
                                int64_t rep_stride, const lhn_pw_opts* opts, const lhn_bnsum* bns, void* stream) {
  if (nrep < 1) nrep = 1;
  lhn_bnsum bs;
  memset(&bs, 0, sizeof(bs));
  if (bns && bns->sums) {
    LHN_CHECK_ARG(x && y && bns->save && dx && stride == 1 && x->C <= 128 && y->C <= 128 && x->C * y->C < 64 * 128 && !x->gate && x->table &&
                      bns->coff >= 0 && bns->coff + x->C <= bns->C,
                  "lhn_conv_pw_bwd3: BatchNorm sums ride in the fused kernel only (stride 1, Cin * Cout < 8192, ungated input inside the producer's channels)");
    bs = *bns;
  }
  LHN_CHECK_ARG(lhn_view_ok(x) && w && y && gy && dw && lhn_no_pend(x) && lhn_no_pend(y), "lhn_conv_pw_bwd: bad view / null pointer");
  LHN_CHECK_ARG(stride == 1 || stride == 2, "lhn_conv_pw_bwd: stride %d", stride);
  LHN_CHECK_ARG(stride == 1 || !dx || dx_accumulate, "lhn_conv_pw_bwd: stride-2 dgrad only accumulates into a zeroed gradient");
  if (!dy_nchw) LHN_CHECK_ARG(lhn_view_ok(y) && gy->dz, "lhn_conv_pw_bwd: bad output view / missing dz");
  const int Cout = y->C, Cin = x->C, HoWo = y->H * y->W;
  const int wcols = (opts && opts->w_cols > 0) ? opts->w_cols : Cin, wrows = (opts && opts->w_rows > 0) ? opts->w_rows : Cout;
  LHN_CHECK_ARG(wcols <= Cin && wcols > Cin - 4 && wrows <= Cout, "lhn_conv_pw_bwd: weight [%d][%d] does not fit views %d -> %d",
                wrows, wcols, Cin, Cout);
  const int64_t bstride = (opts && opts->nchw_batch_stride > 0) ? opts->nchw_batch_stride : (int64_t)Cout * HoWo;
  hipStream_t s = (hipStream_t)stream;
  if (!bs.sums && stride == 1 && !dy_nchw && Cin * Cout >= 64 * 128 && Cout % 32 == 0 && Cin % 32 == 0 && Cout <= 256 && wcols == Cin && wrows == Cout) {
    const int rc = lhn_pw_bwd_split(x, w, y, gy, dx, dx_accumulate, dw, dbias, nrep, rep_stride, s);
    if (rc == 0) {
      LHN_CHECK_LAUNCH("lhn_conv_pw_bwd");
      return 0;
    }
    if (rc > 0) return rc;
  }
  for (int co0 = 0; co0 < Cout; co0 += 128) {
    const int cc = Cout - co0 < 128 ? Cout - co0 : 128, nto = (cc + 31) / 32 == 3 ? 4 : (cc + 31) / 32;
    for (int k0 = 0; k0 < Cin; k0 += 128) {
      const int kc = Cin - k0 < 128 ? Cin - k0 : 128, ci = kc <= 32 ? 32 : pw_cin_tile(kc);   // (no 16-wide backward tile)
      lhn_view xv = *x, yv = *y;
      xv.coff += k0;
      xv.C = kc;
      if (!dy_nchw) {
        yv.coff += co0;
        yv.C = cc;
      }
      PwGeom g;
      g.wstride = wcols;
      g.kvalid = wcols - k0 < kc ? wcols - k0 : kc;
      g.wrows = wrows - co0 < cc ? (wrows - co0 < 0 ? 0 : wrows - co0) : cc;
      g.yacc = 0;
      g.statC = Cout;
      g.nchw_bstride = bstride;
      const float* wv = w + (int64_t)co0 * wcols + k0;
      float* dwv = dw + (int64_t)co0 * wcols + k0;
      float* dbv = (dbias && k0 == 0) ? dbias + co0 : nullptr;
      const float* dyv = dy_nchw ? dy_nchw + (int64_t)co0 * HoWo : nullptr;
      const int acc = dx_accumulate || co0 > 0;
      int rc = -1;
#define PWB_CASE(CI, NTV) \
  if (ci == CI && nto == NTV) rc = launch_pw_bwd<CI, NTV>(&xv, wv, &yv, gy, dx, acc, dwv, dbv, stride, dyv, cc, nrep, rep_stride, g, s, bs);
      PWB_CASE(32, 1) PWB_CASE(32, 2) PWB_CASE(32, 4) PWB_CASE(64, 1) PWB_CASE(64, 2) PWB_CASE(64, 4) PWB_CASE(128, 1)
      PWB_CASE(128, 2) PWB_CASE(128, 4)
#undef PWB_CASE
      LHN_CHECK_ARG(rc != -1, "lhn_conv_pw_bwd: unsupported channels Cin=%d Cout=%d", Cin, Cout);
      if (rc) return rc;
    }
  }
  LHN_CHECK_LAUNCH("lhn_conv_pw_bwd");
  return 0;
}

extern "C" int lhn_conv_pw_bwd(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, float* dx,
                               int dx_accumulate, float* dw, float* dbias, int stride, const float* dy_nchw, int nrep,
                               int64_t rep_stride, void* stream) {
  return lhn_conv_pw_bwd2(x, w, y, gy, dx, dx_accumulate, dw, dbias, stride, dy_nchw, nrep, rep_stride, nullptr, stream);
}
