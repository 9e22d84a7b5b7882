// Dense 3x3 convolution (pad 1, stride 1/2) as an fp32-MFMA implicit GEMM, NHWC:
//     Y[m][co] = sum_tap sum_ci X[m shifted by tap][ci] * W[co][ci][tap]
// i.e. nine accumulated shifted pointwise GEMMs (liteHandNet.py:23-54 BasicBlock / BottleNeck, :183 stem).
// MODE 0 forward : A = consumed input value (pending BN/act applied on load), epilogue = raw store + BN statistics
// MODE 1 dgrad   : A = dy formed on the fly from (dz, saved raw y, BN-backward coefficients), B = W^T with the
//                  taps mirrored, epilogue = dx store / accumulate
// wgrad          : one tap per blockIdx.y, dW_tap[co][ci] += dY^T X_shifted in registers across the block's tiles.
// Tile geometry, LDS images (+4 float row pad) and fragment maps are those of k_conv_pw.hip.
#include "lhn_common.h"

template <int KD, int NT, int MODE, int TAPS>
__global__ void __launch_bounds__(256) k_kxk(lhn_view x, const float* __restrict__ w, lhn_view y, lhn_gradview gy,
                                             double* __restrict__ stats, float* __restrict__ dx, int dx_acc, int stride,
                                             int nout, int M, int ntiles, lhn_bnfin fin) {
  // MODE 0: KD = Cin,  nout = Cout, rows = output pixels of y, A from x
  // MODE 1: KD = Cout, nout = Cin,  rows = input pixels of x,  A from (gy, y)
  constexpr int LDA = KD + 4, C4 = KD / 4, RP = 256 / C4, PF = KD / 8;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                   // [32*NT][LDA]  weights of the current tap
  float* As = smem + 32 * NT * LDA;   // [128][LDA]
  float* red = As + 128 * LDA;        // [4][32*NT][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int c4 = tid % C4, row0 = tid / C4;
  const lhn_view& av = MODE == 0 ? x : y;          // view the A operand is read from
  const lhn_view& ov = MODE == 0 ? y : x;          // geometry of the GEMM rows
  const int cabs = av.coff + 4 * c4;
  const Xf4 xf = lhn_load_xf(av, cabs);
  Gr4 gr;
  if (MODE == 1) gr = lhn_load_coef(gy, y.cstride, cabs);
  const int OW = ov.W, OHW = ov.H * ov.W;
  const int cin_total = MODE == 0 ? KD : nout;     // Cin of the OIHW weight
  float ssum[NT], ssq[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) ssum[nt] = ssq[nt] = 0.f;

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int rn[PF], rh[PF], rw[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const int m = tile * 128 + row0 + p * RP;
      if (m < M) {
        rn[p] = m / OHW;
        const int r = m - rn[p] * OHW;
        rh[p] = r / OW;
        rw[p] = r - rh[p] * OW;
      } else {
        rn[p] = -1;
        rh[p] = rw[p] = 0;
      }
    }
    f16v acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

    for (int tap = 0; tap < TAPS; ++tap) {
      const int kh = TAPS == 1 ? 1 : tap / 3, kw = TAPS == 1 ? 1 : tap - (tap / 3) * 3;   // 1x1 = centre tap only
      // ---- weights of this tap -> LDS as [n][k]   (1x1: once per block)
      if (TAPS > 1 || tile == (int)blockIdx.x) {
        if (MODE == 0 || TAPS > 1) {
          for (int i = tid; i < 32 * NT * KD; i += 256) {
            const int nn = i / KD, kk = i - nn * KD;
            float v = 0.f;
            if (nn < nout) {
              if (MODE == 0) v = w[((size_t)nn * cin_total + kk) * TAPS + tap];        // W[co=nn][ci=kk][tap]
              else v = w[((size_t)kk * cin_total + nn) * TAPS + tap];                  // W[co=kk][ci=nn][tap]
            }
            Ws[nn * LDA + kk] = v;
          }
        } else {
          // 1x1 dgrad: W^T; read rows of W[co=kk][ci=nn] coalesced along ci, scatter into the [n][k] image
          for (int i = tid; i < KD * 32 * NT; i += 256) {
            const int kk = i / (32 * NT), nn = i - kk * (32 * NT);
            Ws[nn * LDA + kk] = nn < nout ? w[(size_t)kk * cin_total + nn] : 0.f;
          }
        }
      }
      // ---- A tile of this tap
#pragma unroll
      for (int p = 0; p < PF; ++p) {
        const int row = row0 + p * RP;
        f4 v = (f4){0.f, 0.f, 0.f, 0.f};
        if (rn[p] >= 0) {
          const int n = rn[p];
          if (MODE == 0) {
            const int ih = rh[p] * stride - 1 + kh, iw = rw[p] * stride - 1 + kw;
            if (ih >= 0 && ih < x.H && iw >= 0 && iw < x.W) {
              const size_t off = ((size_t)(n * x.H + ih) * x.W + iw) * x.cstride + cabs;
              v = lhn_apply_xf(*reinterpret_cast<const f4*>(x.data + off), xf);
              if (x.gate) v *= *reinterpret_cast<const f4*>(x.gate + (size_t)n * x.cstride + cabs);
            }
          } else {
            const int hn = rh[p] + 1 - kh, wn = rw[p] + 1 - kw;
            if (hn >= 0 && wn >= 0 && hn % stride == 0 && wn % stride == 0) {
              const int ho = hn / stride, wo = wn / stride;
              if (ho < y.H && wo < y.W) {
                const size_t off = ((size_t)(n * y.H + ho) * y.W + wo) * y.cstride + cabs;
                const f4 raw = *reinterpret_cast<const f4*>(y.data + off);
                const f4 dz = *reinterpret_cast<const f4*>(gy.dz + off);
                const f4 du = lhn_grad_du(y, gy, xf, raw, dz, n, ho, wo, cabs);
                v = gr.A * du + gr.B * raw + gr.Cc;
              }
            }
          }
        }
        *reinterpret_cast<f4*>(As + row * LDA + 4 * c4) = v;
      }
      __syncthreads();
      const float* arow = As + (wave * 32 + l31) * LDA + 4 * lh;
      const float* brow = Ws + l31 * LDA + 4 * lh;
#pragma unroll 4
      for (int kc = 0; kc < KD / 8; ++kc) {
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
      __syncthreads();
    }
    // ---- epilogue
    const int mbase = tile * 128 + wave * 32 + 4 * lh;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int ch = nt * 32 + l31;
      if (ch >= nout) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mbase + (r & 3) + 8 * (r >> 2);
        if (m < M) {
          const float v = acc[nt][r];
          if (MODE == 0) {
            y.data[(size_t)m * y.cstride + y.coff + ch] = v;
            ssum[nt] += v;
            ssq[nt] += v * v;
          } else {
            float* o = dx + (size_t)m * x.cstride + x.coff + ch;
            *o = dx_acc ? *o + v : v;
          }
        }
      }
    }
  }
  if (MODE == 0 && stats) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float s = ssum[nt] + __shfl_xor(ssum[nt], 32, 64);
      const float q = ssq[nt] + __shfl_xor(ssq[nt], 32, 64);
      if (lh == 0) {
        red[(wave * 32 * NT + nt * 32 + l31) * 2 + 0] = s;
        red[(wave * 32 * NT + nt * 32 + l31) * 2 + 1] = q;
      }
    }
    __syncthreads();
    if (tid < 32 * NT && tid < nout) {
      double s = 0, q = 0;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) {
        s += (double)red[(wv * 32 * NT + tid) * 2 + 0];
        q += (double)red[(wv * 32 * NT + tid) * 2 + 1];
      }
      double* st = stats + (size_t)(blockIdx.x % LHN_STAT_REPLICAS) * 2 * nout;
      atomicAdd(st + tid, s);
      atomicAdd(st + nout + tid, q);
    }
    if (fin.counter && lhn_last_block(fin.counter)) lhn_bn_finalize_block(fin, stats);
  }
}

// wgrad: blockIdx.y = tap.  64-pixel tiles; dYs[m][co], Xs[m][ci] (X shifted by the tap) -> dW_tap += dY^T X.
template <int CIN, int NTO, int TAPS>
__global__ void __launch_bounds__(256) k_kxk_wgrad(lhn_view x, lhn_view y, lhn_gradview gy, float* __restrict__ dw, int stride,
                                                   int cout, int M, int ntiles, int nrep, int64_t rep_stride) {
  constexpr int NTI = CIN / 32, COP = 32 * NTO, LDY = COP + 4, LDX = CIN + 4;
  constexpr int NDW = (NTO * NTI + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* dYs = smem;               // [64][LDY]
  float* Xs = dYs + 64 * LDY;      // [64][LDX]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int tap = blockIdx.y, kh = TAPS == 1 ? 1 : tap / 3, kw = TAPS == 1 ? 1 : tap - (tap / 3) * 3;
  constexpr int XC4 = CIN / 4, XRP = 256 / XC4, XPF = 64 / XRP;
  constexpr int YC4 = COP / 4, YRP = 256 / YC4, YPF = 64 / YRP;
  const int xc4 = tid % XC4, xr0 = tid / XC4, xabs = x.coff + 4 * xc4;
  const int yc4 = tid % YC4, yr0 = tid / YC4, yabs = y.coff + 4 * yc4;
  const Xf4 xxf = lhn_load_xf(x, xabs);
  const bool ych_ok = 4 * yc4 < cout;
  Xf4 yxf;
  Gr4 ygr;
  if (ych_ok) {
    yxf = lhn_load_xf(y, yabs);
    ygr = lhn_load_coef(gy, y.cstride, yabs);
  }
  const int HoWo = y.H * y.W;
  f16v accw[NDW];
#pragma unroll
  for (int t = 0; t < NDW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[t][r] = 0.f;

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#pragma unroll
    for (int p = 0; p < XPF; ++p) {
      const int row = xr0 + p * XRP, m = tile * 64 + row;
      f4 v = (f4){0.f, 0.f, 0.f, 0.f};
      if (m < M) {
        const int n = m / HoWo, r = m - n * HoWo, ho = r / y.W, wo = r - ho * y.W;
        const int ih = ho * stride - 1 + kh, iw = wo * stride - 1 + kw;
        if (ih >= 0 && ih < x.H && iw >= 0 && iw < x.W) {
          v = lhn_apply_xf(*reinterpret_cast<const f4*>(x.data + ((size_t)(n * x.H + ih) * x.W + iw) * x.cstride + xabs), xxf);
          if (x.gate) v *= *reinterpret_cast<const f4*>(x.gate + (size_t)n * x.cstride + xabs);
        }
      }
      *reinterpret_cast<f4*>(Xs + row * LDX + 4 * xc4) = v;
    }
#pragma unroll
    for (int p = 0; p < YPF; ++p) {
      const int row = yr0 + p * YRP, m = tile * 64 + row;
      f4 v = (f4){0.f, 0.f, 0.f, 0.f};
      if (m < M && ych_ok) {
        const int n = m / HoWo, r = m - n * HoWo, h = r / y.W, ww = r - h * y.W;
        const f4 raw = *reinterpret_cast<const f4*>(y.data + (size_t)m * y.cstride + yabs);
        const f4 dz = *reinterpret_cast<const f4*>(gy.dz + (size_t)m * y.cstride + yabs);
        const f4 du = lhn_grad_du(y, gy, yxf, raw, dz, n, h, ww, yabs);
        v = ygr.A * du + ygr.B * raw + ygr.Cc;
      }
      *reinterpret_cast<f4*>(dYs + row * LDY + 4 * yc4) = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int ks = 0; ks < 32; ++ks) {
      const float* dyr = dYs + (2 * ks + lh) * LDY + l31;
      const float* xr = Xs + (2 * ks + lh) * LDX + l31;
#pragma unroll
      for (int t = 0; t < NDW; ++t) {
        const int tl = wave + 4 * t;
        if (tl < NTO * NTI) {
          const int it = tl / NTI, jt = tl % NTI;
          accw[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(dyr[32 * it], xr[32 * jt], accw[t], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  dw += (size_t)(blockIdx.x % nrep) * rep_stride;
#pragma unroll
  for (int t = 0; t < NDW; ++t) {
    const int tl = wave + 4 * t;
    if (tl < NTO * NTI) {
      const int it = tl / NTI, jt = tl % NTI;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = 32 * it + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (co < cout) atomicAdd(dw + ((size_t)co * CIN + 32 * jt + l31) * TAPS + tap, accw[t][r]);
      }
    }
  }
}

template <int KD, int NT, int MODE, int TAPS>
static int launch_kxk(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, double* stats, float* dx,
                      int dx_acc, int stride, int nout, hipStream_t s, const lhn_bnfin* finp = nullptr) {
  lhn_bnfin fin;
  if (finp && stats) fin = *finp; else fin.counter = nullptr;
  const lhn_view* ov = MODE == 0 ? y : x;
  const int M = ov->N * ov->H * ov->W, ntiles = (M + 127) / 128;
  const size_t lds = (size_t)((32 * NT + 128) * (KD + 4) + 4 * 32 * NT * 2) * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_kxk<KD, NT, MODE, TAPS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) {
      lhn_set_error("lhn_conv_kxk: cannot reserve %zu B of LDS", lds);
      return 2;
    }
    attr_done = true;
  }
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu > 4) per_cu = 4;
  if (per_cu < 1) per_cu = 1;
  int grid = lhn_num_cus() * per_cu;
  if (grid > ntiles) grid = ntiles;
  lhn_gradview g;
  if (gy) g = *gy; else g.dz = g.dpool = g.coef = nullptr;
  hipLaunchKernelGGL((k_kxk<KD, NT, MODE, TAPS>), dim3(grid), dim3(256), lds, s, *x, w, *y, g, stats, dx, dx_acc, stride, nout, M, ntiles, fin);
  return 0;
}

template <int CIN, int NTO, int TAPS>
static int launch_kxk_wgrad(const lhn_view* x, const lhn_view* y, const lhn_gradview* gy, float* dw, int stride, int nrep,
                            int64_t rep_stride, hipStream_t s) {
  const int M = y->N * y->H * y->W, ntiles = (M + 63) / 64;
  constexpr int COP = 32 * NTO;
  const size_t lds = (size_t)(64 * (COP + 4) + 64 * (CIN + 4)) * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_kxk_wgrad<CIN, NTO, TAPS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) {
      lhn_set_error("lhn_conv_kxk_bwd: cannot reserve %zu B of LDS", lds);
      return 2;
    }
    attr_done = true;
  }
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu > 3) per_cu = 3;
  if (per_cu < 1) per_cu = 1;
  int grid = TAPS == 1 ? lhn_num_cus() * per_cu : lhn_num_cus() / 4;      // x 9 taps
  if (grid < 1) grid = 1;
  if (grid > ntiles) grid = ntiles;
  hipLaunchKernelGGL((k_kxk_wgrad<CIN, NTO, TAPS>), dim3(grid, TAPS), dim3(256), lds, s, *x, *y, *gy, dw, stride, y->C, M, ntiles, nrep,
                     rep_stride);
  return 0;
}

static int kxk_geometry_ok(const lhn_view* x, const lhn_view* y, int stride) {
  return y->N == x->N && y->H == (x->H - 1) / stride + 1 && y->W == (x->W - 1) / stride + 1;
}

extern "C" int lhn_conv_kxk_fwd(const lhn_view* x, const float* w, const lhn_view* y, double* stats, int stride,
                                const lhn_bnfin* fin, void* stream) {
  LHN_CHECK_ARG(lhn_view_ok(x) && lhn_view_ok(y) && w, "lhn_conv_kxk_fwd: bad view / null pointer");
  LHN_CHECK_ARG((stride == 1 || stride == 2) && kxk_geometry_ok(x, y, stride), "lhn_conv_kxk_fwd: geometry / stride %d", stride);
  const int nt = (y->C + 31) / 32;
  hipStream_t s = (hipStream_t)stream;
  int rc = -1;
#define KF(CI, NTV) if (x->C == CI && nt == NTV) rc = launch_kxk<CI, NTV, 0, 9>(x, w, y, nullptr, stats, nullptr, 0, stride, y->C, s, fin);
  KF(32, 1) KF(64, 2) KF(128, 4) KF(32, 2) KF(64, 1) KF(64, 4) KF(128, 2) KF(128, 1) KF(32, 4)
#undef KF
  LHN_CHECK_ARG(rc != -1, "lhn_conv_kxk_fwd: unsupported channels Cin=%d Cout=%d", x->C, y->C);
  if (rc) return rc;
  LHN_CHECK_LAUNCH("lhn_conv_kxk_fwd");
  return 0;
}

extern "C" int lhn_conv_kxk_bwd(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, float* dx,
                                int dx_accumulate, float* dw, int stride, int nrep, int64_t rep_stride, void* stream) {
  LHN_CHECK_ARG(lhn_view_ok(x) && lhn_view_ok(y) && w && gy && gy->dz && dw, "lhn_conv_kxk_bwd: bad view / null pointer");
  LHN_CHECK_ARG((stride == 1 || stride == 2) && kxk_geometry_ok(x, y, stride), "lhn_conv_kxk_bwd: geometry / stride %d", stride);
  if (nrep < 1) nrep = 1;
  hipStream_t s = (hipStream_t)stream;
  int rc = -1;
  if (dx) {
    const int nt = (x->C + 31) / 32;   // GEMM N = Cin, K = Cout
#define KB(CO, NTV) if (y->C == CO && nt == NTV) rc = launch_kxk<CO, NTV, 1, 9>(x, w, y, gy, nullptr, dx, dx_accumulate, stride, x->C, s);
    KB(32, 1) KB(64, 2) KB(128, 4) KB(32, 2) KB(64, 1) KB(64, 4) KB(128, 2) KB(128, 1) KB(32, 4)
#undef KB
    LHN_CHECK_ARG(rc != -1, "lhn_conv_kxk_bwd: unsupported channels Cin=%d Cout=%d", x->C, y->C);
    if (rc) return rc;
  }
  rc = -1;
  const int nto = (y->C + 31) / 32;
#define KW(CI, NTV) if (x->C == CI && nto == NTV) rc = launch_kxk_wgrad<CI, NTV, 9>(x, y, gy, dw, stride, nrep, rep_stride, s);
  KW(32, 1) KW(64, 2) KW(128, 4) KW(32, 2) KW(64, 1) KW(64, 4) KW(128, 2) KW(128, 1) KW(32, 4)
#undef KW
  LHN_CHECK_ARG(rc != -1, "lhn_conv_kxk_bwd: unsupported channels Cin=%d Cout=%d", x->C, y->C);
  if (rc) return rc;
  LHN_CHECK_LAUNCH("lhn_conv_kxk_bwd");
  return 0;
}

// 1x1 backward as two launches (dgrad = the implicit-GEMM kernel with one tap, wgrad = the per-tap kernel with one
// tap): used by lhn_conv_pw_bwd for large Cin*Cout where the fused kernel is LDS-bound to one block per CU.
int lhn_pw_bwd_split(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, float* dx, int dx_accumulate,
                     float* dw, int nrep, int64_t rep_stride, hipStream_t s) {
  int rc = -1;
  if (dx) {
    const int nt = (x->C + 31) / 32;
#define PB(CO, NTV) if (y->C == CO && nt == NTV) rc = launch_kxk<CO, NTV, 1, 1>(x, w, y, gy, nullptr, dx, dx_accumulate, 1, x->C, s);
    PB(64, 4) PB(128, 4) PB(128, 2) PB(64, 2) PB(128, 1) PB(32, 4)
#undef PB
    if (rc) return rc;
  }
  rc = -1;
  const int nto = (y->C + 31) / 32;
#define PW(CI, NTV) if (x->C == CI && nto == NTV) rc = launch_kxk_wgrad<CI, NTV, 1>(x, y, gy, dw, 1, nrep, rep_stride, s);
  PW(64, 4) PW(128, 4) PW(128, 2) PW(64, 2) PW(128, 1) PW(32, 4)
#undef PW
  return rc;
}
