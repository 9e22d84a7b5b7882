// Dense 3x3 convolution (pad 1, stride 1/2) as an fp32-MFMA implicit GEMM, NHWC:
//     Y[m][co] = sum_tap sum_ci X[m shifted by tap][ci] * W[co][ci][tap]
// i.e. nine accumulated shifted pointwise GEMMs (liteHandNet.py:23-54 BasicBlock / BottleNeck, :183 stem).
// MODE 0 forward : A = consumed input value (pending BN/act applied on load), epilogue = raw store + BN statistics
// MODE 1 dgrad   : A = dy formed on the fly from (dz, saved raw y, BN-backward coefficients), B = W^T with the
//                  taps mirrored, epilogue = dx store / accumulate
// wgrad          : one tap per blockIdx.y, dW_tap[co][ci] += dY^T X_shifted in registers across the block's tiles.
// Tile geometry, LDS images (+4 float row pad) and fragment maps are those of k_conv_pw.hip.
#include <stdlib.h>
#include "lhn_common.h"

// dz := dy in place (dy = A*du + B*y + C with du from gate / pooled gradient / leaky derivative).  Used ahead of the
// MFMA-heavy backward kernels: they then stream ONE plain tensor instead of (dz, raw y) + the formula per element,
// which halves their prefetch registers (no spills) -- worth one extra elementwise pass when Cin*Cout is large.
__global__ void __launch_bounds__(256) k_dy_inplace(lhn_view y, lhn_gradview g, float* __restrict__ dz, float* __restrict__ dbias,
                                                    int nrep, int64_t rep_stride) {
  __shared__ f4 red[256];
  const int C4 = y.C >> 2, c4 = threadIdx.x % C4, pl = threadIdx.x / C4, PL = 256 / C4;
  const int ca = y.coff + 4 * c4;
  const Xf4 xf = lhn_load_xf(y, ca);
  const Gr4 gr = lhn_load_coef(g, y.cstride, ca);
  const int rows = y.N * y.H;
  f4 bsum = (f4){0.f, 0.f, 0.f, 0.f};
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const int n = row / y.H, h = row - n * y.H;
    for (int w = LHN_LANE0(pl, PL); w < y.W; w += PL) {
      const size_t off = ((size_t)row * y.W + w) * y.cstride + ca;
      const f4 raw = *reinterpret_cast<const f4*>(y.data + off);
      const f4 du = lhn_grad_du(y, g, xf, raw, *reinterpret_cast<const f4*>(dz + off), n, h, w, ca);
      const f4 dy = gr.A * du + gr.B * raw + gr.Cc;
      *reinterpret_cast<f4*>(dz + off) = dy;
      bsum += dy;
    }
  }
  // d(bias) of a biased convolution = column sums of dy: this pass already touches every element.  One thread per CHANNEL
  // adds its column (consecutive lanes -> consecutive addresses: the full-rate atomic form) into this block's gradient replica.
  if (dbias) {
    red[threadIdx.x] = bsum;                 // (lanes beyond PL * C4 never entered the loop: zero)
    __syncthreads();
    if ((int)threadIdx.x < y.C) {
      const float* rf = reinterpret_cast<const float*>(red);
      const int cc = threadIdx.x >> 2, jj = threadIdx.x & 3;
      float v = 0.f;
      for (int j = 0; j < PL; ++j) v += rf[(j * C4 + cc) * 4 + jj];
      atomicAdd(dbias + (size_t)(blockIdx.x % nrep) * rep_stride + threadIdx.x, v);
    }
  }
}
static void launch_dy_inplace(const lhn_view* y, const lhn_gradview* gy, hipStream_t s, float* dbias = nullptr, int nrep = 1,
                              int64_t rep_stride = 0) {
  int64_t g = (int64_t)y->N * y->H;
  const int64_t cap = (int64_t)lhn_num_cus() * 8;
  if (g > cap) g = cap;
  hipLaunchKernelGGL(k_dy_inplace, dim3((int)g), dim3(256), 0, s, *y, *gy, const_cast<float*>(gy->dz), dbias, nrep < 1 ? 1 : nrep, rep_stride);
}

// ---------------------------------------------------------------------------------------------------------
// Core.  GEMM rows = 128 pixels per tile (wave w owns rows [32w, 32w+32)), N = 32*NT features, one PHASE =
// one tap with the whole K = KD: 16*NT*KD/32... MFMAs per wave, long enough to cover memory latency with a single
// 4-wave workgroup per CU.  While a phase multiplies out of LDS, the A rows of the NEXT phase are already in
// flight into registers (raw loads; the pending BN/activation/gate or the dy formula is applied when they are
// committed to LDS after the MFMA loop).  Weights of the next tap are fetched at the start of the commit; for
// 1x1 (TAPS = 1) they are staged once per block.
template <int KD, int NT, int MODE, int TAPS, bool PLAIN = false>
__global__ void __launch_bounds__(256) k_kxk(lhn_view x, const float* __restrict__ w, lhn_view y, lhn_gradview gy,
                                             double* __restrict__ stats, float* __restrict__ dx, int dx_acc, int stride,
                                             int nout, int Mhost, int ntiles, lhn_bnfin fin, const float* __restrict__ wt) {
  constexpr int BM = 128, LDA = KD + 4;
  constexpr int C4 = KD / 4, RP = 256 / C4, PF = BM / RP;     // float4 per thread per phase
  constexpr int NW = 32 * NT * KD / 256;                      // weight scalars per thread per tap
  constexpr int WCH = NW > 16 ? 16 : NW;                      // staged in chunks of <= 16 registers
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                   // [32*NT][LDA]
  float* As = smem + 32 * NT * LDA;   // [128][LDA]
  float* red = As + BM * LDA;         // [4][32*NT][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lh = lane >> 5;
  const int c4 = tid % C4, row0 = tid / C4;
  const int n0 = blockIdx.y * 32 * NT;        // N split: this block's first output feature (small M: more blocks)
  const lhn_view& av = MODE == 0 ? x : y;
  const lhn_view& ov = MODE == 0 ? y : x;
  // Stride-2 dgrad by input-pixel PARITY (gridDim.z == 4): an input pixel (ih, iw) only meets the taps with
  // kh = ih+1 (mod 2), kw = iw+1 (mod 2) -- 1, 2, 2 or 4 of the 9 -- so each parity class is its own small implicit GEMM
  // over its quarter of the pixels and its own tap list (2.25 taps per pixel on average instead of 9, 3/4 of them zeros).
  const bool par = (MODE == 1 && TAPS == 9 && gridDim.z == 4);
  const int ph = par ? (int)(blockIdx.z >> 1) : 0, pw2 = par ? (int)(blockIdx.z & 1) : 0;
  const int OH2 = par ? (ov.H - ph + 1) / 2 : ov.H, OW = par ? (ov.W - pw2 + 1) / 2 : ov.W, OHW = OH2 * OW;
  const int M = par ? ov.N * OHW : Mhost;
  const int nkw = (par && !pw2) ? 1 : (par ? 2 : 3), ntaps = par ? (ph ? 2 : 1) * nkw : TAPS;
  auto tap_of = [&](int t) __attribute__((always_inline)) -> int {
    if (!par) return t;
    const int a = t / nkw, b = t - a * nkw;
    return (ph ? 2 * a : 1) * 3 + (pw2 ? 2 * b : 1);
  };
  const int cin_total = MODE == 0 ? KD : nout;
  const int cabs = av.coff + 4 * c4;
  const Xf4 xf = lhn_load_xf(av, cabs);
  Gr4 gr;
  if (MODE == 1) gr = lhn_load_coef(gy, y.cstride, cabs);
  double ssum[NT], ssq[NT];          // per-tile fp32 partials promoted to double (see k_conv_pw.hip)
#pragma unroll
  for (int j = 0; j < NT; ++j) ssum[j] = ssq[j] = 0.0;

  constexpr bool LIN = (TAPS == 1);        // 1x1: launched with stride 1 only -> rows are linear pixel indices
  int rn[LIN ? 1 : PF], rh[LIN ? 1 : PF], rw[LIN ? 1 : PF];
  f4 pa[PF], pb[(MODE == 1 && !PLAIN) ? PF : 1];
  bool pv[PF];
  int cur_tile = 0;

  auto geom = [&](int tile) __attribute__((always_inline)) {
    cur_tile = tile;
    if (LIN) return;
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      // branch-free (an if/else over these small arrays made the compiler keep them in scratch memory)
      const int m = tile * BM + row0 + p * RP, mc = min(m, M - 1);
      const int n = mc / OHW, r = mc - n * OHW, i2 = r / OW, j2 = r - i2 * OW;
      rn[p] = m < M ? n : -1;
      rh[p] = par ? 2 * i2 + ph : i2;
      rw[p] = par ? 2 * j2 + pw2 : j2;
    }
  };
  auto issue = [&](int tap) __attribute__((always_inline)) {
    const int kh = TAPS == 1 ? 1 : tap / 3, kw = TAPS == 1 ? 1 : tap - (tap / 3) * 3;
    if (LIN) {
#pragma unroll
      for (int p = 0; p < PF; ++p) {
        const int m = cur_tile * BM + row0 + p * RP;
        pv[p] = m < M;
        const size_t off = (size_t)min(m, M - 1) * av.cstride + cabs;
        if (MODE == 1 && PLAIN) pa[p] = *reinterpret_cast<const f4*>(gy.dz + off);
        else pa[p] = *reinterpret_cast<const f4*>(av.data + off);
        if (MODE == 1 && !PLAIN) pb[(MODE == 1 && !PLAIN) ? p : 0] = *reinterpret_cast<const f4*>(gy.dz + off);
      }
      return;
    }
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      // clamped, always-valid addresses: all loads of the phase issue back to back; validity is a select at commit
      const int n = rn[p] < 0 ? 0 : rn[p];
      if (MODE == 0) {
        const int ih = rh[p] * stride - 1 + kh, iw = rw[p] * stride - 1 + kw;
        const int ihc = min(max(ih, 0), x.H - 1), iwc = min(max(iw, 0), x.W - 1);
        pv[p] = rn[p] >= 0 && ih == ihc && iw == iwc;
        pa[p] = *reinterpret_cast<const f4*>(x.data + ((size_t)(n * x.H + ihc) * x.W + iwc) * x.cstride + cabs);
      } else {
        const int hn = rh[p] + 1 - kh, wn2 = rw[p] + 1 - kw;
        const int ho = hn / stride, wo = wn2 / stride;     // hn, wn2 >= -1: trunc == floor where it matters (validity below)
        const int hoc = min(max(ho, 0), y.H - 1), woc = min(max(wo, 0), y.W - 1);
        pv[p] = rn[p] >= 0 && hn >= 0 && wn2 >= 0 && ho * stride == hn && wo * stride == wn2 && ho == hoc && wo == woc;
        const size_t off = ((size_t)(n * y.H + hoc) * y.W + woc) * y.cstride + cabs;
        if (PLAIN) pa[p] = *reinterpret_cast<const f4*>(gy.dz + off);
        else {
          pa[p] = *reinterpret_cast<const f4*>(y.data + off);
          pb[(MODE == 1 && !PLAIN) ? p : 0] = *reinterpret_cast<const f4*>(gy.dz + off);
        }
      }
    }
  };
  auto stage_w = [&](int tap) __attribute__((always_inline)) {
    if (TAPS > 1 && wt) {
      // tap-major copy of the weights (k_w_tapmajor): row = output feature of this GEMM, KD contiguous floats per row ->
      // 16-byte loads/stores instead of 4-byte gathers with a 36-byte stride (which cost more than the MFMAs of a phase)
      constexpr int NWV = 32 * NT * KD / 4 / 256, K4 = KD / 4;
#pragma unroll
      for (int j0 = 0; j0 < NWV; j0 += 4) {
        f4 t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = tid + 256 * (j0 + j), nn = i / K4, k4 = i - nn * K4;
          t[j] = *reinterpret_cast<const f4*>(wt + ((size_t)tap * nout + min(n0 + nn, nout - 1)) * KD + 4 * k4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = tid + 256 * (j0 + j), nn = i / K4, k4 = i - nn * K4;
          *reinterpret_cast<f4*>(Ws + nn * LDA + 4 * k4) = n0 + nn < nout ? t[j] : (f4){0.f, 0.f, 0.f, 0.f};
        }
      }
      return;
    }
#pragma unroll 1
    for (int j0 = 0; j0 < NW; j0 += WCH) {
      float t[WCH];
#pragma unroll
      for (int j = 0; j < WCH; ++j) {
        const int i = tid + 256 * (j0 + j);
        float v = 0.f;
        if (MODE == 0 || TAPS > 1) {
          const int nn = i / KD, kk = i - nn * KD;
          if (n0 + nn < nout) v = MODE == 0 ? w[((size_t)(n0 + nn) * cin_total + kk) * TAPS + tap] : w[((size_t)kk * cin_total + n0 + nn) * TAPS + tap];
        } else {   // 1x1 dgrad: W[co = k][ci = n], coalesced along n
          const int kk = i / (32 * NT), nn = i - kk * (32 * NT);
          if (n0 + nn < nout) v = w[(size_t)kk * cin_total + n0 + nn];
        }
        t[j] = v;
      }
#pragma unroll
      for (int j = 0; j < WCH; ++j) {
        const int i = tid + 256 * (j0 + j);
        if (MODE == 0 || TAPS > 1) {
          const int nn = i / KD, kk = i - nn * KD;
          Ws[nn * LDA + kk] = t[j];
        } else {
          const int kk = i / (32 * NT), nn = i - kk * (32 * NT);
          Ws[nn * LDA + kk] = t[j];
        }
      }
    }
  };
  auto commit = [&](int tap) __attribute__((always_inline)) {
    const int kh = TAPS == 1 ? 1 : tap / 3, kw = TAPS == 1 ? 1 : tap - (tap / 3) * 3;
    const f4 one = (f4){1.f, 1.f, 1.f, 1.f}, zero = (f4){0.f, 0.f, 0.f, 0.f};
    if (MODE == 1 && PLAIN) {
#pragma unroll
      for (int p = 0; p < PF; ++p) *reinterpret_cast<f4*>(As + (row0 + p * RP) * LDA + 4 * c4) = pv[p] ? pa[p] : zero;
      return;
    }
    const float* gptr = av.gate;
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      int n = 0;
      if (gptr) n = LIN ? min(cur_tile * BM + row0 + p * RP, M - 1) / OHW : (rn[p] < 0 ? 0 : rn[p]);
      const f4 gate = gptr ? *reinterpret_cast<const f4*>(gptr + (size_t)n * av.cstride + cabs) : one;
      f4 v;
      if (MODE == 0) v = lhn_apply_xf(pa[p], xf) * gate;
      else v = lhn_dy_fast(xf, gr, pa[p], pb[(MODE == 1 && !PLAIN) ? p : 0], gate);
      *reinterpret_cast<f4*>(As + (row0 + p * RP) * LDA + 4 * c4) = pv[p] ? v : zero;
    }
  };

  int tile = blockIdx.x;
  if (tile < ntiles) {
    geom(tile);
    issue(tap_of(0));
    stage_w(tap_of(0));
    commit(tap_of(0));
  }
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    f16v acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int next_tile = tile + gridDim.x;
    for (int ti = 0; ti < ntaps; ++ti) {
      const bool more = (ti + 1 < ntaps) || (next_tile < ntiles);
      const int ntap = tap_of(ti + 1 < ntaps ? ti + 1 : 0);
      if (ti + 1 < ntaps) issue(ntap);
      else if (next_tile < ntiles) {
        geom(next_tile);
        issue(ntap);
      }
      const float* arow = As + (wave * 32 + l31) * LDA + 4 * lh;
      const float* brow = Ws + l31 * LDA + 4 * lh;
#pragma unroll 4
      for (int kc = 0; kc < KD / 8; ++kc) {
        const f4 a = *reinterpret_cast<const f4*>(arow + kc * 8);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const f4 b = *reinterpret_cast<const f4*>(brow + j * 32 * LDA + kc * 8);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[j], 0, 0, 0);
        }
      }
      if (more) {
        __syncthreads();                 // every wave is done reading this phase's LDS images
        if (TAPS > 1) stage_w(ntap);
        commit(ntap);
        __syncthreads();
      }
    }
    // ---- epilogue (C/D layout: col = lane&31 -> feature, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -> pixel)
    const int mbase = tile * BM + wave * 32 + 4 * lh;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int ch = n0 + j * 32 + l31;
      if (ch >= nout) continue;
      TileStat ts;
      ts.reset();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mbase + (r & 3) + 8 * (r >> 2);
        if (m < M) {
          const float v = acc[j][r];
          if (MODE == 0) {
            y.data[(size_t)m * y.cstride + y.coff + ch] = v;
            ts.add(v);
          } else {
            size_t pix = (size_t)m;
            if (par) {
              const int n = m / OHW, rr = m - n * OHW, i2 = rr / OW, j2 = rr - i2 * OW;
              pix = ((size_t)n * x.H + 2 * i2 + ph) * x.W + 2 * j2 + pw2;
            }
            float* o = dx + pix * x.cstride + x.coff + ch;
            *o = dx_acc ? *o + v : v;
          }
        }
      }
      if (MODE == 0) ts.flush(ssum[j], ssq[j]);
    }
  }
  if (MODE == 0 && stats) {
    __syncthreads();
    double* redd = reinterpret_cast<double*>(As);        // [4][32*NT][2] doubles inside the (now idle) A tile
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const double s = ssum[j] + __shfl_xor(ssum[j], 32, 64);
      const double q = ssq[j] + __shfl_xor(ssq[j], 32, 64);
      if (lh == 0) {
        redd[(wave * 32 * NT + j * 32 + l31) * 2 + 0] = s;
        redd[(wave * 32 * NT + j * 32 + l31) * 2 + 1] = q;
      }
    }
    __syncthreads();
    if (tid < 32 * NT && n0 + tid < nout) {
      double s = 0, q = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        s += redd[(k * 32 * NT + tid) * 2 + 0];
        q += redd[(k * 32 * NT + tid) * 2 + 1];
      }
      double* st = stats + (size_t)(blockIdx.x % LHN_STAT_REPLICAS) * 2 * nout;
      atomicAdd(st + n0 + tid, s);
      atomicAdd(st + nout + n0 + tid, q);
    }
    if (fin.counter && lhn_last_block(fin.counter)) lhn_bn_finalize_block(fin, stats);
  }
}

// wgrad: blockIdx.y = tap, blockIdx.z = group of 32*NTO output channels (co0).  When gridDim.x <= nrep every (pixel
// chunk, tap, co group) owns a private slice of gradient replica blockIdx.x and the flush is a plain read-add-write: the
// float atomics of the flush (gridDim.x * |dW| of them, ~38 G/s) were the whole cost of this kernel on small maps.
// wgrad: blockIdx.y = tap.  64-pixel tiles; dYs[m][co], Xs[m][ci] (X shifted by the tap) -> dW_tap += dY^T X.
template <int CIN, int NTO, int TAPS, bool PLAIN = false>
__global__ void __launch_bounds__(256) k_kxk_wgrad(lhn_view x, lhn_view y, lhn_gradview gy, float* __restrict__ dw, int stride,
                                                   int cout, int M, int ntiles, int nrep, int64_t rep_stride, int wstride) {
  constexpr int NTI = CIN / 32, COP = 32 * NTO, LDY = COP + 4, LDX = CIN + 4;
  // fewer dW tiles than waves (32 -> 32: one tile): split the 64-pixel K range of a tile over the idle waves; the partial
  // tiles meet in LDS before the flush
  constexpr int T = NTO * NTI, KS = T < 4 ? 4 / T : 1;
  constexpr int NDW = (T * KS + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* dYs = smem;               // [64][LDY]
  float* Xs = dYs + 64 * LDY;      // [64][LDX]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lh = lane >> 5;
  const int tap = blockIdx.y, kh = TAPS == 1 ? 1 : tap / 3, kw = TAPS == 1 ? 1 : tap - (tap / 3) * 3;
  constexpr int XC4 = CIN / 4, XRP = 256 / XC4, XPF = 64 / XRP;
  constexpr int YC4 = COP / 4, YRP = 256 / YC4, YPF = 64 / YRP;
  const int xc4 = tid % XC4, xr0 = tid / XC4, xabs = x.coff + 4 * xc4;
  const int co0 = blockIdx.z * COP;
  const int yc4 = tid % YC4, yr0 = tid / YC4, yabs = y.coff + co0 + 4 * yc4;
  const Xf4 xxf = lhn_load_xf(x, xabs);
  const bool ych_ok = co0 + 4 * yc4 < cout;
  const Xf4 yxf = lhn_load_xf(y, ych_ok ? yabs : y.coff);
  const Gr4 ygr = lhn_load_coef(gy, y.cstride, ych_ok ? yabs : y.coff);
  const int HoWo = y.H * y.W;
  f16v accw[NDW];
#pragma unroll
  for (int t = 0; t < NDW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[t][r] = 0.f;

  const f4 one = (f4){1.f, 1.f, 1.f, 1.f}, zero = (f4){0.f, 0.f, 0.f, 0.f};
  f4 xraw[XPF], yraw[PLAIN ? 1 : YPF], ydz[YPF];
  bool xok[XPF], yok[YPF];
  int xn[XPF], yn[YPF];
  // raw global loads of one 64-pixel tile into registers (clamped addresses, validity kept as predicates)
  auto issue = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < XPF; ++p) {
      const int m = min(tile * 64 + xr0 + p * XRP, M - 1);
      const int n = m / HoWo, r = m - n * HoWo, ho = r / y.W, wo = r - ho * y.W;
      const int ih = ho * stride - 1 + kh, iw = wo * stride - 1 + kw;
      const int ihc = min(max(ih, 0), x.H - 1), iwc = min(max(iw, 0), x.W - 1);
      xok[p] = (tile * 64 + xr0 + p * XRP < M) && ih == ihc && iw == iwc;
      xn[p] = n;
      xraw[p] = *reinterpret_cast<const f4*>(x.data + ((size_t)(n * x.H + ihc) * x.W + iwc) * x.cstride + xabs);
    }
#pragma unroll
    for (int p = 0; p < YPF; ++p) {
      const int m = min(tile * 64 + yr0 + p * YRP, M - 1);
      yok[p] = (tile * 64 + yr0 + p * YRP < M) && ych_ok;
      yn[p] = m / HoWo;
      const size_t off = (size_t)m * y.cstride + (ych_ok ? yabs : y.coff);
      if (!PLAIN) yraw[PLAIN ? 0 : p] = *reinterpret_cast<const f4*>(y.data + off);
      ydz[p] = *reinterpret_cast<const f4*>(gy.dz + off);
    }
  };
  // transform and park them in LDS
  auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < XPF; ++p) {
      const f4 gate = x.gate ? *reinterpret_cast<const f4*>(x.gate + (size_t)xn[p] * x.cstride + xabs) : one;
      const f4 v = lhn_apply_xf(xraw[p], xxf) * gate;
      *reinterpret_cast<f4*>(Xs + (xr0 + p * XRP) * LDX + 4 * xc4) = xok[p] ? v : zero;
    }
    if (PLAIN) {
#pragma unroll
      for (int p = 0; p < YPF; ++p) *reinterpret_cast<f4*>(dYs + (yr0 + p * YRP) * LDY + 4 * yc4) = yok[p] ? ydz[p] : zero;
    } else {
#pragma unroll
      for (int p = 0; p < YPF; ++p) {
        const f4 gate = (y.gate && ych_ok) ? *reinterpret_cast<const f4*>(y.gate + (size_t)yn[p] * y.cstride + yabs) : one;
        const f4 v = lhn_dy_fast(yxf, ygr, yraw[PLAIN ? 0 : p], ydz[p], gate);
        *reinterpret_cast<f4*>(dYs + (yr0 + p * YRP) * LDY + 4 * yc4) = yok[p] ? v : zero;
      }
    }
  };
  // The next tile's raw loads are issued before this tile's MFMA phase and parked in LDS after it (PF).  Only where every
  // wave owns whole dW tiles: the K-split instances (NTO*NTI < 4) produced wrong dW in this form on hipcc 7.2 and keep the
  // plain issue -> commit -> multiply order.
  constexpr bool PF = (KS == 1);
  int tile = blockIdx.x;
  if (PF && tile < ntiles) issue(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    if (!PF) issue(tile);
    commit();
    __syncthreads();
    if (PF && tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x);
    if constexpr (KS == 1 && T % 4 == 0 && 4 % NTI == 0) {
      // whole tiles per wave: NO predicate around the MFMAs and the wave index in a scalar register.  With `if (tl < T)` in this
      // loop the compiler emitted exec-mask branches and an lgkmcnt(0) wait in front of every MFMA group: 128 -> 128 3x3 at
      // 64 x 64 took 840 us (58 % of the fp32 MFMA peak); this form: hourglass step 66.4 -> 64.3 ms, A 19.87 -> 19.39.  (A variant
      // with the three taps of a kernel row per workgroup -- one dY tile, a 66-pixel X run, 12 accumulator tiles per wave, one
      // workgroup per CU -- was built on top and measured 0.1-1 % SLOWER than one tap per workgroup: removed.)
      const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
      const float* ap = dYs + lh * LDY + l31 + 32 * (wv / NTI);
      const float* bp = Xs + lh * LDX + l31 + 32 * (wv % NTI);
#pragma unroll 4
      for (int ks = 0; ks < 32; ++ks) {
        const float b = bp[(2 * ks) * LDX];
        float a[NDW];
#pragma unroll
        for (int t = 0; t < NDW; ++t) a[t] = ap[(2 * ks) * LDY + 32 * (4 / NTI) * t];
#pragma unroll
        for (int t = 0; t < NDW; ++t) accw[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b, accw[t], 0, 0, 0);
      }
    } else {
      const int kpart = KS > 1 ? wave % KS : 0;
#pragma unroll 4
      for (int ks = kpart * (32 / KS); ks < (kpart + 1) * (32 / KS); ++ks) {
        const float* dyr = dYs + (2 * ks + lh) * LDY + l31;
        const float* xr = Xs + (2 * ks + lh) * LDX + l31;
#pragma unroll
        for (int t = 0; t < NDW; ++t) {
          const int tl = (wave + 4 * t) / KS;
          if (tl < T) {
            const int it = tl / NTI, jt = tl % NTI;
            accw[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(dyr[32 * it], xr[32 * jt], accw[t], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
  }
  const bool exclusive = (int)gridDim.x <= nrep;
  dw += (size_t)(blockIdx.x % nrep) * rep_stride;
  if (KS > 1) {
    // K-split: NDW == 1; waves of a tile (wave / KS equal) add their partials through LDS (dYs/Xs are dead), the first
    // wave of each tile keeps the sum
    float* part = dYs;                        // [4 waves][16 regs][64 lanes]
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) part[(wave * 16 + r) * 64 + lane] = accw[0][r];
    __syncthreads();
    if (wave % KS == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < KS; ++k) v += part[((wave + k) * 16 + r) * 64 + lane];
        accw[0][r] = v;
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NDW; ++t) {
    const int tl = (wave + 4 * t) / KS;
    if (tl < T && (KS == 1 || wave % KS == 0)) {
      const int it = tl / NTI, jt = tl % NTI;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + 32 * it + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (co < cout) {
          float* o = dw + ((size_t)co * wstride + 32 * jt + l31) * TAPS + tap;     // wstride = input channels of the WHOLE weight
          if (exclusive) *o += accw[t][r];
          else atomicAdd(o, accw[t][r]);
        }
      }
    }
  }
}

// wt[tap][r][k]: mode 0 (forward GEMM, rows = Cout, K = Cin)  wt = w[r][k][tap];  mode 1 (dgrad, rows = Cin, K = Cout)
// wt = w[k][r][tap].  147,456 floats for 128 -> 128: a ~3 us launch per conv and direction.
__global__ void __launch_bounds__(256) k_w_tapmajor(const float* __restrict__ w, float* __restrict__ wt, int Cout, int Cin, int mode) {
  const int R = mode ? Cin : Cout, K = mode ? Cout : Cin, total = 9 * R * K;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int k = i % K, r = (i / K) % R, tap = i / (K * R);
    wt[i] = mode ? w[((size_t)k * Cin + r) * 9 + tap] : w[((size_t)r * Cin + k) * 9 + tap];
  }
}
static void launch_w_tapmajor(const float* w, float* wt, int Cout, int Cin, int mode, hipStream_t s) {
  const int total = 9 * Cout * Cin;
  hipLaunchKernelGGL(k_w_tapmajor, dim3((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024), dim3(256), 0, s, w, wt, Cout, Cin, mode);
}

template <int KD, int NT, int MODE, int TAPS, bool PLAIN = false>
static int launch_kxk(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, double* stats, float* dx,
                      int dx_acc, int stride, int nout, hipStream_t s, const lhn_bnfin* finp = nullptr, int nsplit = 1, int parity = 0,
                      const float* wt = nullptr) {
  lhn_bnfin fin;
  if (finp && stats) fin = *finp; else fin.counter = nullptr;
  const lhn_view* ov = MODE == 0 ? y : x;
  const int M = parity ? ov->N * ((ov->H + 1) / 2) * ((ov->W + 1) / 2) : ov->N * ov->H * ov->W, ntiles = (M + 127) / 128;
  const size_t lds = (size_t)((32 * NT + 128) * (KD + 4) + 4 * 32 * NT * 2) * sizeof(float);
  static LhnKernelCfg cfg;
  int per_cu = 1;
  if (!lhn_kernel_cfg(cfg, &k_kxk<KD, NT, MODE, TAPS, PLAIN>, lds, 4, &per_cu)) {
    lhn_set_error("lhn_conv_kxk: cannot reserve %zu B of LDS", lds);
    return 2;
  }
  int grid = lhn_num_cus() * per_cu;
  if (grid > ntiles) grid = ntiles;
  lhn_gradview g;
  if (gy) g = *gy; else g.dz = g.dpool = g.coef = nullptr;
  hipLaunchKernelGGL((k_kxk<KD, NT, MODE, TAPS, PLAIN>), dim3(grid, nsplit, parity ? 4 : 1), dim3(256), lds, s, *x, w, *y, g, stats, dx, dx_acc, stride, nout, M, ntiles, fin, wt);
  return 0;
}

template <int CIN, int NTO, int TAPS, bool PLAIN = false>
static int launch_kxk_wgrad(const lhn_view* x, const lhn_view* y, const lhn_gradview* gy, float* dw, int stride, int nrep,
                            int64_t rep_stride, hipStream_t s, int cosplit = 1, int wstride = CIN) {
  const int M = y->N * y->H * y->W, ntiles = (M + 63) / 64;
  constexpr int COP = 32 * NTO;
  const size_t lds = (size_t)(64 * (COP + 4) + 64 * (CIN + 4)) * sizeof(float);
  static LhnKernelCfg cfg;
  int per_cu = 1;
  if (!lhn_kernel_cfg(cfg, &k_kxk_wgrad<CIN, NTO, TAPS, PLAIN>, lds, 3, &per_cu)) {
    lhn_set_error("lhn_conv_kxk_bwd: cannot reserve %zu B of LDS", lds);
    return 2;
  }
  int grid = TAPS == 1 ? lhn_num_cus() * per_cu / cosplit : lhn_num_cus() / 4;      // x 9 taps
  if (TAPS > 1 && cosplit > 1 && nrep > 1) grid = nrep;                   // exclusive replica slices: no atomics in the flush
  if (grid < 1) grid = 1;
  if (grid > ntiles) grid = ntiles;
  hipLaunchKernelGGL((k_kxk_wgrad<CIN, NTO, TAPS, PLAIN>), dim3(grid, TAPS, cosplit), dim3(256), lds, s, *x, *y, *gy, dw, stride, y->C, M, ntiles, nrep,
                     rep_stride, wstride);
  return 0;
}

static int kxk_geometry_ok(const lhn_view* x, const lhn_view* y, int stride) {
  return y->N == x->N && y->H == (x->H - 1) / stride + 1 && y->W == (x->W - 1) / stride + 1;
}

// Small maps give few 128-pixel tiles (8x8 maps at batch 64: 32 tiles for 256 CUs): split the N = 32*nt output features
// over gridDim.y so that about two workgroups per CU exist.  Returns the per-block tile count (1, 2 or 4 -> nt / splits).
static int kxk_nt_block(int M, int nt) {
  const int ntiles = (M + 127) / 128;
  int splits = 1;
  while (splits < nt && nt % (splits * 2) == 0 && ntiles * splits < 2 * lhn_num_cus()) splits *= 2;
  return nt / splits;
}

extern "C" int lhn_conv_kxk_fwd(const lhn_view* x, const float* w, const lhn_view* y, double* stats, int stride,
                                const lhn_bnfin* fin, float* wt_scratch, void* stream) {
  LHN_CHECK_ARG(lhn_view_ok(x) && lhn_view_ok(y) && w && lhn_no_pend(x), "lhn_conv_kxk_fwd: bad view / null pointer (no pending BatchNorm here)");
  LHN_CHECK_ARG((stride == 1 || stride == 2) && kxk_geometry_ok(x, y, stride), "lhn_conv_kxk_fwd: geometry / stride %d", stride);
  const int ntot = (y->C + 31) / 32, nt = (ntot == 1 || ntot == 2 || ntot == 4) ? kxk_nt_block(y->N * y->H * y->W, ntot) : ntot;
  hipStream_t s = (hipStream_t)stream;
  int rc = -1;
  if (wt_scratch) launch_w_tapmajor(w, wt_scratch, y->C, x->C, 0, s);
#define KF(CI, NTV) if (x->C == CI && nt == NTV) rc = launch_kxk<CI, NTV, 0, 9>(x, w, y, nullptr, stats, nullptr, 0, stride, y->C, s, fin, ntot / nt, 0, wt_scratch);
  KF(32, 1) KF(64, 2) KF(128, 4) KF(32, 2) KF(64, 1) KF(64, 4) KF(128, 2) KF(128, 1) KF(32, 4)
#undef KF
  LHN_CHECK_ARG(rc != -1, "lhn_conv_kxk_fwd: unsupported channels Cin=%d Cout=%d", x->C, y->C);
  if (rc) return rc;
  LHN_CHECK_LAUNCH("lhn_conv_kxk_fwd");
  return 0;
}

extern "C" int lhn_conv_kxk_bwd(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, float* dx,
                                int dx_accumulate, float* dw, int stride, int nrep, int64_t rep_stride, float* wt_scratch,
                                void* stream) {
  LHN_CHECK_ARG(lhn_view_ok(x) && lhn_view_ok(y) && w && gy && gy->dz && dw && lhn_no_pend(x) && lhn_no_pend(y), "lhn_conv_kxk_bwd: bad view / null pointer");
  LHN_CHECK_ARG((stride == 1 || stride == 2) && kxk_geometry_ok(x, y, stride), "lhn_conv_kxk_bwd: geometry / stride %d", stride);
  if (nrep < 1) nrep = 1;
  hipStream_t s = (hipStream_t)stream;
  int rc = -1;
  // large channel counts (or a channel-attention pooled gradient): turn dz into dy in place, then stream it plain
  const char* pe = getenv("LHN_PLAIN");
  const bool plain = pe ? (pe[0] == '1') : ((x->C * y->C >= 64 * 64) || gy->dpool);
  if (plain) launch_dy_inplace(y, gy, s);
  if (dx) {
    // GEMM N = Cin, K = Cout
    const int par = stride == 2 ? 1 : 0;     // stride 2: four parity classes of input pixels (gridDim.z)
    if (wt_scratch) launch_w_tapmajor(w, wt_scratch, y->C, x->C, 1, s);
    const int ntot = (x->C + 31) / 32, nt = (ntot == 1 || ntot == 2 || ntot == 4) ? kxk_nt_block(x->N * x->H * x->W, ntot) : ntot;
#define KB(CO, NTV) if (y->C == CO && nt == NTV) rc = plain ? launch_kxk<CO, NTV, 1, 9, true>(x, w, y, gy, nullptr, dx, dx_accumulate, stride, x->C, s, nullptr, ntot / nt, par, wt_scratch) : launch_kxk<CO, NTV, 1, 9, false>(x, w, y, gy, nullptr, dx, dx_accumulate, stride, x->C, s, nullptr, ntot / nt, par, wt_scratch);
    KB(32, 1) KB(64, 2) KB(128, 4) KB(32, 2) KB(64, 1) KB(64, 4) KB(128, 2) KB(128, 1) KB(32, 4)
#undef KB
    LHN_CHECK_ARG(rc != -1, "lhn_conv_kxk_bwd: unsupported channels Cin=%d Cout=%d", x->C, y->C);
    if (rc) return rc;
  }
  rc = -1;
  // Cout >= 64: one 32-channel group per block (gridDim.z), pixel chunks = gradient replicas -> atomic-free flush
  // (only while a pixel chunk stays short: <= 16 tiles of 64 pixels per block; big maps amortise the atomic flush)
  const int ntot = (y->C + 31) / 32, wtiles = (y->N * y->H * y->W + 63) / 64;
  const int cosplit = (ntot == 2 || ntot == 4) && x->C >= 64 && nrep > 1 && wtiles <= 16 * nrep ? ntot : 1, nto = ntot / cosplit;
#define KW(CI, NTV) if (x->C == CI && nto == NTV) rc = plain ? launch_kxk_wgrad<CI, NTV, 9, true>(x, y, gy, dw, stride, nrep, rep_stride, s, cosplit) : launch_kxk_wgrad<CI, NTV, 9, false>(x, y, gy, dw, stride, nrep, rep_stride, s, cosplit);
  KW(32, 1) KW(64, 2) KW(128, 4) KW(32, 2) KW(64, 1) KW(64, 4) KW(128, 2) KW(128, 1) KW(32, 4)
#undef KW
  LHN_CHECK_ARG(rc != -1, "lhn_conv_kxk_bwd: unsupported channels Cin=%d Cout=%d", x->C, y->C);
  if (rc) return rc;
  LHN_CHECK_LAUNCH("lhn_conv_kxk_bwd");
  return 0;
}

// 1x1 backward as two launches (dgrad = the implicit-GEMM kernel with one tap, wgrad = the per-tap kernel with one
// tap): used by lhn_conv_pw_bwd for large Cin*Cout where the fused kernel is LDS-bound to one block per CU.  Channel counts
// above 128 (hourglass: 256) run as 128-wide slices: dgrad accumulates over output-channel slices (its K), wgrad runs once
// per input-channel slice with the output channels on gridDim.z.  Returns -1 when a shape has no instance (the caller then
// takes the fused kernel).
int lhn_pw_dgrad_wr(const lhn_view* dyv, const float* w, const lhn_view* dxv, int wstride, int accumulate, hipStream_t s);

int lhn_pw_bwd_split(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, float* dx, int dx_accumulate,
                     float* dw, float* dbias, int nrep, int64_t rep_stride, hipStream_t s) {
  const int Cin = x->C, Cout = y->C;
  auto dgrad_ok = [&](int co, int nt) { return (co == 64 && (nt == 4 || nt == 2)) || (co == 128 && (nt == 4 || nt == 2 || nt == 1)) || (co == 32 && nt == 4); };
  auto wgrad_ok = [&](int ci, int nto) { return (ci == 64 && (nto == 4 || nto == 2)) || (ci == 128 && (nto == 4 || nto == 2 || nto == 1)) || (ci == 32 && nto == 4); };
  // every slice must have an instance BEFORE anything is launched (dy is formed in place: no way back afterwards)
  const int ntd = Cin >= 128 ? 4 : (Cin + 31) / 32;                  // dgrad: features per block = 32*ntd, the rest on gridDim.y
  if (Cin > 128 && Cin % 128 != 0) return -1;
  for (int co0 = 0; co0 < Cout; co0 += 128)
    if (dx && !dgrad_ok(Cout - co0 < 128 ? Cout - co0 : 128, ntd)) return -1;
  const int nto = Cout >= 128 ? 4 : (Cout + 31) / 32;
  if (Cout > 128 && Cout % 128 != 0) return -1;
  for (int k0 = 0; k0 < Cin; k0 += 128)
    if (!wgrad_ok(Cin - k0 < 128 ? Cin - k0 : 128, nto)) return -1;
  launch_dy_inplace(y, gy, s, dbias, nrep, rep_stride);
  int rc = -1;
  bool dgrad_done = false;
  if (dx && Cout == 256 && x->C % 32 == 0) {
    // K = 256 in one pass on the register-W kernel (dy as input, W transposed): no second K slice accumulating into dx
    lhn_view dyv = *y;
    dyv.data = const_cast<float*>(gy->dz);
    dyv.table = nullptr;
    dyv.gate = nullptr;
    dyv.pend = nullptr;
    dgrad_done = true;
    for (int ci0 = 0; ci0 < Cin && dgrad_done; ci0 += 128) {
      lhn_view dxv = *x;
      dxv.data = dx;
      dxv.table = nullptr;
      dxv.gate = nullptr;
      dxv.pend = nullptr;
      dxv.coff = x->coff + ci0;
      dxv.C = Cin - ci0 < 128 ? Cin - ci0 : 128;
      const int r2 = lhn_pw_dgrad_wr(&dyv, w + ci0, &dxv, Cin, dx_accumulate, s);
      if (r2 > 0) return r2;
      if (r2 < 0) {
        if (ci0 > 0) return 3;
        dgrad_done = false;
      }
    }
  }
  if (dx && !dgrad_done) {
    for (int co0 = 0; co0 < Cout; co0 += 128) {
      const int cc = Cout - co0 < 128 ? Cout - co0 : 128;
      lhn_view yv = *y;
      yv.coff += co0;
      yv.C = cc;
      const float* wv = w + (size_t)co0 * Cin;
      const int acc = dx_accumulate || co0 > 0, nsplit = (Cin + 32 * ntd - 1) / (32 * ntd);
      // the register-resident-weights 1x1 kernel (k_conv_pw.hip) run on dy with W transposed, <= 128 input channels per launch
      if ((cc == 32 || cc == 64 || cc == 128) && x->C % 32 == 0) {
        lhn_view dyv = yv;
        dyv.data = const_cast<float*>(gy->dz);
        dyv.table = nullptr;
        dyv.gate = nullptr;
        dyv.pend = nullptr;
        bool ok = true;
        for (int ci0 = 0; ci0 < Cin && ok; ci0 += 128) {
          lhn_view dxv = *x;
          dxv.data = dx;
          dxv.table = nullptr;
          dxv.gate = nullptr;
          dxv.pend = nullptr;
          dxv.coff = x->coff + ci0;
          dxv.C = Cin - ci0 < 128 ? Cin - ci0 : 128;
          const int r2 = lhn_pw_dgrad_wr(&dyv, wv + ci0, &dxv, Cin, acc, s);
          if (r2 > 0) return r2;
          if (r2 < 0) {
            if (ci0 > 0) return 3;      // (cannot happen: the first slice decides whether there is an instance)
            ok = false;
          }
        }
        if (ok) continue;
      }
      rc = -1;
#define PB(CO, NTV) if (cc == CO && ntd == NTV) rc = launch_kxk<CO, NTV, 1, 1, true>(x, wv, &yv, gy, nullptr, dx, acc, 1, x->C, s, nullptr, nsplit);
      PB(64, 4) PB(128, 4) PB(128, 2) PB(64, 2) PB(128, 1) PB(32, 4)
#undef PB
      if (rc) return rc < 0 ? 3 : rc;
    }
  }
  for (int k0 = 0; k0 < Cin; k0 += 128) {
    const int kc = Cin - k0 < 128 ? Cin - k0 : 128;
    lhn_view xv = *x;
    xv.coff += k0;
    xv.C = kc;
    const int cosplit = (Cout + 32 * nto - 1) / (32 * nto);
    rc = -1;
#define PW(CI, NTV) if (kc == CI && nto == NTV) rc = launch_kxk_wgrad<CI, NTV, 1, true>(&xv, y, gy, dw + k0, 1, nrep, rep_stride, s, cosplit, Cin);
    PW(64, 4) PW(128, 4) PW(128, 2) PW(64, 2) PW(128, 1) PW(32, 4)
#undef PW
    if (rc) return rc < 0 ? 3 : rc;
  }
  return 0;
}
