// Attention of `mynet` (reference models/pose_hg_ms_att.py:165-174, 191-192):
//   gate = sigmoid( Linear( dropout( dw3x3_valid( relu( BN( adaptive_avg_pool(y, 3x3) ) ) ) + b3 ) ) ),   y *= gate
// The 3x3 pooling itself is lhn_avgpool_fwd; these kernels run on the pooled [N][9][C] tensor (a few hundred KB).
// BatchNorm statistics are over the N*9 pooled values of a channel.
// save layout (floats): ad[N*C] (dropout output = Linear input) | g[N*C] | mean[C] | invstd[C] | dad[N*C] (backward scratch)
#include "lhn_common.h"

// grid = ceil(C/32) blocks; thread = (channel lane 0..31, sample lane 0..NL-1), NL = blockDim.x / 32 (32 as launched)
__global__ void __launch_bounds__(1024) k_att1(const float* __restrict__ pooled, const float* __restrict__ gamma,
                                              const float* __restrict__ beta, float* __restrict__ rmean,
                                              float* __restrict__ rvar, int64_t* __restrict__ nbt,
                                              const float* __restrict__ w3, const float* __restrict__ b3,
                                              const float* __restrict__ mask, float* __restrict__ save, int N, int C, float eps,
                                              float momentum, int training, int stage, double* __restrict__ gsum,
                                              double count_scale) {
  __shared__ double rs[32][32], rq[32][32];
  const int NL = blockDim.x >> 5;
  __shared__ float s_mean[32], s_inv[32];
  float* ad = save;
  float* smean = save + (int64_t)N * C * 2;
  float* sinv = smean + C;
  const int cl = threadIdx.x & 31, nl = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
  const bool ok = c < C;
  double s = 0, q = 0;
  if (ok && training && stage != 2)
    for (int n = nl; n < N; n += NL)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float p = pooled[((int64_t)n * 9 + t) * C + c];
        s += p;
        q += (double)p * p;
      }
  rs[nl][cl] = s;
  rq[nl][cl] = q;
  __syncthreads();
  if (nl == 0 && ok) {
    for (int j = 1; j < NL; ++j) {
      s += rs[j][cl];
      q += rq[j][cl];
    }
    if (stage == 1) {
      gsum[c] = s;
      gsum[C + c] = q;
    } else if (stage == 2) {
      s = gsum[c];
      q = gsum[C + c];
    }
  }
  if (stage == 1) return;
  if (nl == 0 && ok) {
    double mean, var;
    if (training) {
      const double cnt = 9.0 * N * count_scale;
      mean = s / cnt;
      var = q / cnt - mean * mean;
      if (var < 0) var = 0;
      rmean[c] = (float)((1.0 - (double)momentum) * rmean[c] + (double)momentum * mean);
      rvar[c] = (float)((1.0 - (double)momentum) * rvar[c] + (double)momentum * (cnt > 1 ? var * cnt / (cnt - 1.0) : var));
    } else {
      mean = rmean[c];
      var = rvar[c];
    }
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    smean[c] = s_mean[cl] = (float)mean;
    sinv[c] = s_inv[cl] = invstd;
  }
  __syncthreads();
  if (ok) {
    const float mean = s_mean[cl], sc = gamma[c] * s_inv[cl], sh = beta[c];
    float wt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t] = w3[c * 9 + t];
    const float bb = b3 ? b3[c] : 0.f;
    for (int n = nl; n < N; n += NL) {
      float a = bb;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float u = (pooled[((int64_t)n * 9 + t) * C + c] - mean) * sc + sh;
        a += wt[t] * fmaxf(u, 0.f);
      }
      if (mask) a *= mask[(int64_t)n * C + c];
      ad[(int64_t)n * C + c] = a;
    }
  }
  if (training && nbt && blockIdx.x == 0 && threadIdx.x == 0) nbt[0] += 1;
}

// one block per sample: gate[n][co] = sigmoid(bl[co] + sum_c wl[co][c] * ad[n][c])
__global__ void __launch_bounds__(256) k_att2(const float* __restrict__ wl, const float* __restrict__ bl,
                                              float* __restrict__ save, float* __restrict__ gate, int gs, int gcoff, int N,
                                              int C) {
  __shared__ float sa[256];
  const int n = blockIdx.x;
  const float* ad = save + (int64_t)n * C;
  float* g = save + (int64_t)N * C + (int64_t)n * C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) sa[c] = ad[c];
  __syncthreads();
  for (int co = threadIdx.x; co < C; co += blockDim.x) {
    float v = bl[co];
    for (int c = 0; c < C; ++c) v += wl[co * C + c] * sa[c];
    const float sg = 1.f / (1.f + expf(-v));
    g[co] = sg;
    gate[(int64_t)n * gs + gcoff + co] = sg;
  }
}

// backward of the Linear + sigmoid: dad[n][c], dwl, dbl
__global__ void __launch_bounds__(256) k_att_bwd2(const float* __restrict__ wl, const float* __restrict__ save,
                                                  const float* __restrict__ dgate, float* __restrict__ dad,
                                                  float* __restrict__ dwl, float* __restrict__ dbl, int N, int C) {
  __shared__ float sdz[256], sa[256];
  // one workgroup per sample; the deterministic mode launches ONE workgroup that walks the samples in order (the parameter
  // gradients are sums over samples into one destination: every address then has a single writer and a fixed order)
  for (int n = blockIdx.x; n < N; n += gridDim.x) {
    __syncthreads();
    const float* ad = save + (int64_t)n * C;
    const float* g = save + (int64_t)N * C + (int64_t)n * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      const float gg = g[c];
      const float dz = dgate[(int64_t)n * C + c] * gg * (1.f - gg);
      sdz[c] = dz;
      sa[c] = ad[c];
      atomicAdd(dbl + c, dz);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) atomicAdd(dwl + i, sdz[i / C] * sa[i % C]);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      float d = 0.f;
#pragma unroll 8
      for (int co = 0; co < C; ++co) d += wl[co * C + c] * sdz[co];
      dad[(int64_t)n * C + c] = d;
    }
  }
}

// backward of dropout, dw3x3, relu, BatchNorm; writes the 25-segment pooled gradient (lhn_dpool_store)
__global__ void __launch_bounds__(1024) k_att_bwd1(const float* __restrict__ pooled, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, const float* __restrict__ w3,
                                                  const float* __restrict__ mask, const float* __restrict__ save,
                                                  const float* __restrict__ dad, float* __restrict__ dpool, int cs, int coff,
                                                  int H, int W, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                  float* __restrict__ dw3, float* __restrict__ db3, int N, int C,
                                                  int training, int stage, double* __restrict__ gsum, double count_scale,
                                                  float pgrad_scale) {
  __shared__ double rs[32][32], rq[32][32];
  const int NL = blockDim.x >> 5;
  __shared__ float rw[32][32][10];
  const float* smean = save + (int64_t)N * C * 2;
  const float* sinv = smean + C;
  const int cl = threadIdx.x & 31, nl = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
  const bool ok = c < C;
  const float mean = ok ? smean[c] : 0.f, invstd = ok ? sinv[c] : 0.f, gm = ok ? gamma[c] : 0.f, bt = ok ? beta[c] : 0.f;
  float wt[9], dwt[9], binv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    wt[t] = ok ? w3[c * 9 + t] : 0.f;
    dwt[t] = 0.f;
    const int bi = t / 3, bj = t % 3;
    binv[t] = 1.f / (float)((lhn_bin_hi(bi, H) - lhn_bin_lo(bi, H)) * (lhn_bin_hi(bj, W) - lhn_bin_lo(bj, W)));
  }
  double sd = 0, sdx = 0;
  float dbias = 0.f;
  if (ok)
    for (int n = nl; n < N; n += NL) {
      float da = dad[(int64_t)n * C + c];
      if (mask) da *= mask[(int64_t)n * C + c];
      dbias += da;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float xh = (pooled[((int64_t)n * 9 + t) * C + c] - mean) * invstd;
        const float u = xh * gm + bt;
        const float r = fmaxf(u, 0.f);
        dwt[t] += da * r;
        const float du = u > 0.f ? da * wt[t] : 0.f;
        sd += du;
        sdx += (double)du * xh;
      }
    }
  rs[nl][cl] = sd;
  rq[nl][cl] = sdx;
#pragma unroll
  for (int t = 0; t < 9; ++t) rw[nl][cl][t] = dwt[t];
  rw[nl][cl][9] = dbias;
  __syncthreads();
  sd = 0;
  sdx = 0;
  for (int j = 0; j < NL; ++j) {
    sd += rs[j][cl];
    sdx += rq[j][cl];
  }
  // stage 1 (SyncBatchNorm) stops after the local sums; the local parameter gradients of the conv are final already
  if (nl == 0 && ok && stage != 2) {
    for (int t = 0; t < 10; ++t) {
      float v = 0.f;
      for (int j = 0; j < NL; ++j) v += rw[j][cl][t];
      if (t < 9) dw3[c * 9 + t] += v;
      else if (db3) db3[c] += v;
    }
  }
  if (stage == 1) {
    if (nl == 0 && ok) {
      gsum[c] = sd;
      gsum[C + c] = sdx;
    }
    return;
  }
  if (stage == 2 && ok) {
    sd = gsum[c];
    sdx = gsum[C + c];
  }
  if (nl == 0 && ok) {
    dgamma[c] += (float)sdx * pgrad_scale;
    dbeta[c] += (float)sd * pgrad_scale;
  }
  if (ok) {
    const double cnt = 9.0 * N * count_scale;
    const float m1 = (float)(sd / cnt), m2 = (float)(sdx / cnt);
    for (int n = nl; n < N; n += NL) {
      float da = dad[(int64_t)n * C + c];
      if (mask) da *= mask[(int64_t)n * C + c];
      float dseg[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float xh = (pooled[((int64_t)n * 9 + t) * C + c] - mean) * invstd;
        const float u = xh * gm + bt;
        const float du = u > 0.f ? da * wt[t] : 0.f;
        const float dp = training ? gm * invstd * (du - m1 - xh * m2) : gm * invstd * du;
        dseg[t] = dp * binv[t];
      }
      lhn_dpool_store(dpool, n, cs, coff + c, dseg);
    }
  }
}

extern "C" int lhn_att_mlp_fwd(const float* pooled, const float* gamma, const float* beta, float* rmean, float* rvar, int64_t* nbt,
                               const float* w3, const float* b3, const float* wl, const float* bl, const float* dropmask,
                               float* gate, int gate_stride, int gate_coff, float* save, int N, int C, float eps, float momentum,
                               int training, int stage, double* gsum, double count_scale, void* stream) {
  LHN_CHECK_ARG(pooled && gamma && beta && rmean && rvar && w3 && wl && bl && gate && save, "lhn_att_mlp_fwd: null pointer");
  LHN_CHECK_ARG(C > 0 && C <= 256 && N > 0, "lhn_att_mlp_fwd: C=%d (<=256)", C);
  hipStream_t s = (hipStream_t)stream;
  LHN_CHECK_ARG(stage == 0 || (gsum && stage >= 1 && stage <= 2 && count_scale >= 1), "lhn_att_mlp_fwd: stage %d needs gsum", stage);
  hipLaunchKernelGGL(k_att1, dim3((C + 31) / 32), dim3(1024), 0, s, pooled, gamma, beta, rmean, rvar, nbt, w3, b3, dropmask, save, N,
                     C, eps, momentum, training, stage, gsum, stage ? count_scale : 1.0);
  if (stage != 1) hipLaunchKernelGGL(k_att2, dim3(N), dim3(128), 0, s, wl, bl, save, gate, gate_stride, gate_coff, N, C);
  LHN_CHECK_LAUNCH("lhn_att_mlp_fwd");
  return 0;
}

extern "C" int lhn_att_mlp_bwd(const float* pooled, const float* gamma, const float* beta, const float* w3, const float* wl,
                               const float* dropmask, float* save, const float* dgate, float* dpool, int cstride, int coff, int H,
                               int W, float* dgamma, float* dbeta, float* dw3, float* db3, float* dwl, float* dbl, int N, int C,
                               int stage, double* gsum, double count_scale, float pgrad_scale, void* stream) {
  LHN_CHECK_ARG(pooled && gamma && beta && w3 && wl && save && dgate && dpool && dgamma && dbeta && dw3 && dwl && dbl,
                "lhn_att_mlp_bwd: null pointer");
  LHN_CHECK_ARG(C > 0 && C <= 256 && N > 0, "lhn_att_mlp_bwd: C=%d (<=256)", C);
  hipStream_t s = (hipStream_t)stream;
  float* dad = save + (int64_t)N * C * 2 + 2 * C;
  LHN_CHECK_ARG(stage == 0 || (gsum && stage >= 1 && stage <= 2 && count_scale >= 1), "lhn_att_mlp_bwd: stage %d needs gsum", stage);
  if (stage != 2) hipLaunchKernelGGL(k_att_bwd2, dim3(lhn_deterministic_mode() ? 1 : N), dim3(256), 0, s, wl, save, dgate, dad, dwl, dbl, N, C);
  hipLaunchKernelGGL(k_att_bwd1, dim3((C + 31) / 32), dim3(1024), 0, s, pooled, gamma, beta, w3, dropmask, save, dad, dpool, cstride,
                     coff, H, W, dgamma, dbeta, dw3, db3, N, C, 1, stage, gsum, stage ? count_scale : 1.0, stage ? pgrad_scale : 1.f);
  LHN_CHECK_LAUNCH("lhn_att_mlp_bwd");
  return 0;
}

// ------------------------------------------------------------------ squeeze-and-excitation (reference common.py:23-37)
//   gate = sigmoid(up(relu(down(global_avg_pool(y)))))   -- 1x1 convs with bias, J = internal neurons (C/16)
// pooled = lhn_avgpool_fwd(y, 1, 1) -> [N][C].  save layout (floats): h[N*J] | g[N*C]
__global__ void __launch_bounds__(256) k_se_fwd(const float* __restrict__ pooled, const float* __restrict__ w1,
                                                const float* __restrict__ b1, const float* __restrict__ w2,
                                                const float* __restrict__ b2, float* __restrict__ save, float* __restrict__ gate,
                                                int gs, int gcoff, int N, int C, int J, int mode) {
  // mode 0: SEBlock (common.py:23-37) relu inside, sigmoid outside; mode 1: SpatialWeighting (lite_hrnet.py:55-74):
  // sigmoid(relu(.)) after BOTH 1x1 convolutions
  __shared__ float sp[256], sh[64];
  const int n = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) sp[c] = pooled[(int64_t)n * C + c];
  __syncthreads();
  for (int j = threadIdx.x; j < J; j += blockDim.x) {
    float v = b1[j];
    for (int c = 0; c < C; ++c) v += w1[j * C + c] * sp[c];
    v = fmaxf(v, 0.f);
    if (mode == 1) v = 1.f / (1.f + expf(-v));
    sh[j] = v;
    save[(int64_t)n * J + j] = v;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float v = b2[c];
    for (int j = 0; j < J; ++j) v += w2[c * J + j] * sh[j];
    const float g = 1.f / (1.f + expf(-(mode == 1 ? fmaxf(v, 0.f) : v)));
    save[(int64_t)N * J + (int64_t)n * C + c] = g;
    gate[(int64_t)n * gs + gcoff + c] = g;
  }
}

// backward: parameter gradients (atomics over the N blocks) and the pooled gradient, written to all 25 segment slots
// (one global bin: every pixel receives d(loss)/d(pooled) / (H*W))
__global__ void __launch_bounds__(256) k_se_bwd(const float* __restrict__ pooled, const float* __restrict__ w1,
                                                const float* __restrict__ w2, const float* __restrict__ save,
                                                const float* __restrict__ dgate, float* __restrict__ dpool, int cs, int coff,
                                                float inv_hw, float* __restrict__ dw1, float* __restrict__ db1,
                                                float* __restrict__ dw2, float* __restrict__ db2, int N, int C, int J, int mode) {
  // mode 1: sigmoid(relu(v)) has derivative g(1-g) where v > 0, i.e. where g > 1/2, and 0 elsewhere (both layers)
  __shared__ float sp[256], sh[64], sdz[256], sdh[64];
  for (int n = blockIdx.x; n < N; n += gridDim.x) {      // (deterministic mode: one workgroup, samples in order; see k_att_bwd2)
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    sp[c] = pooled[(int64_t)n * C + c];
    const float g = save[(int64_t)N * J + (int64_t)n * C + c];
    float dz = dgate[(int64_t)n * C + c] * g * (1.f - g);
    if (mode == 1 && !(g > 0.5f)) dz = 0.f;
    sdz[c] = dz;
    atomicAdd(db2 + c, dz);
  }
  for (int j = threadIdx.x; j < J; j += blockDim.x) sh[j] = save[(int64_t)n * J + j];
  __syncthreads();
  for (int i = threadIdx.x; i < C * J; i += blockDim.x) atomicAdd(dw2 + i, sdz[i / J] * sh[i % J]);
  for (int j = threadIdx.x; j < J; j += blockDim.x) {
    float d = 0.f;
#pragma unroll 8
    for (int c = 0; c < C; ++c) d += w2[c * J + j] * sdz[c];
    if (mode == 1) d = sh[j] > 0.5f ? d * sh[j] * (1.f - sh[j]) : 0.f;
    else d = sh[j] > 0.f ? d : 0.f;
    sdh[j] = d;
    atomicAdd(db1 + j, d);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < J * C; i += blockDim.x) atomicAdd(dw1 + i, sdh[i / C] * sp[i % C]);
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float d = 0.f;
#pragma unroll 8
    for (int j = 0; j < J; ++j) d += w1[j * C + c] * sdh[j];
    d *= inv_hw;
    for (int s = 0; s < LHN_DPOOL_SLOTS; ++s) dpool[((int64_t)n * LHN_DPOOL_SLOTS + s) * cs + coff + c] = d;
  }
  }
}

extern "C" int lhn_se_mlp_fwd2(const float* pooled, const float* w1, const float* b1, const float* w2, const float* b2, float* gate,
                               int gate_stride, int gate_coff, float* save, int N, int C, int J, int mode, void* stream);
extern "C" int lhn_se_mlp_fwd(const float* pooled, const float* w1, const float* b1, const float* w2, const float* b2, float* gate,
                              int gate_stride, int gate_coff, float* save, int N, int C, int J, void* stream) {
  return lhn_se_mlp_fwd2(pooled, w1, b1, w2, b2, gate, gate_stride, gate_coff, save, N, C, J, 0, stream);
}
extern "C" int lhn_se_mlp_fwd2(const float* pooled, const float* w1, const float* b1, const float* w2, const float* b2, float* gate,
                               int gate_stride, int gate_coff, float* save, int N, int C, int J, int mode, void* stream) {
  LHN_CHECK_ARG(pooled && w1 && b1 && w2 && b2 && gate && save, "lhn_se_mlp_fwd: null pointer");
  LHN_CHECK_ARG(C > 0 && C <= 256 && J > 0 && J <= 64 && N > 0, "lhn_se_mlp_fwd: C=%d J=%d (C <= 256, J <= 64)", C, J);
  hipLaunchKernelGGL(k_se_fwd, dim3(N), dim3(128), 0, (hipStream_t)stream, pooled, w1, b1, w2, b2, save, gate, gate_stride, gate_coff,
                     N, C, J, mode);
  LHN_CHECK_LAUNCH("lhn_se_mlp_fwd");
  return 0;
}

extern "C" int lhn_se_mlp_bwd2(const float* pooled, const float* w1, const float* w2, const float* save, const float* dgate,
                               float* dpool, int cstride, int coff, int H, int W, float* dw1, float* db1, float* dw2, float* db2,
                               int N, int C, int J, int mode, void* stream);
extern "C" int lhn_se_mlp_bwd(const float* pooled, const float* w1, const float* w2, const float* save, const float* dgate,
                              float* dpool, int cstride, int coff, int H, int W, float* dw1, float* db1, float* dw2, float* db2,
                              int N, int C, int J, void* stream) {
  return lhn_se_mlp_bwd2(pooled, w1, w2, save, dgate, dpool, cstride, coff, H, W, dw1, db1, dw2, db2, N, C, J, 0, stream);
}
extern "C" int lhn_se_mlp_bwd2(const float* pooled, const float* w1, const float* w2, const float* save, const float* dgate,
                               float* dpool, int cstride, int coff, int H, int W, float* dw1, float* db1, float* dw2, float* db2,
                               int N, int C, int J, int mode, void* stream) {
  LHN_CHECK_ARG(pooled && w1 && w2 && save && dgate && dpool && dw1 && db1 && dw2 && db2, "lhn_se_mlp_bwd: null pointer");
  LHN_CHECK_ARG(C > 0 && C <= 256 && J > 0 && J <= 64 && N > 0 && H > 0 && W > 0, "lhn_se_mlp_bwd: C=%d J=%d", C, J);
  hipLaunchKernelGGL(k_se_bwd, dim3(lhn_deterministic_mode() ? 1 : N), dim3(256), 0, (hipStream_t)stream, pooled, w1, w2, save, dgate, dpool, cstride, coff,
                     1.f / (float)(H * W), dw1, db1, dw2, db2, N, C, J, mode);
  LHN_CHECK_LAUNCH("lhn_se_mlp_bwd");
  return 0;
}
