// BatchNorm finalize, elementwise combine (+nearest upsample), 2x2 max-pool, adaptive average pool,
// channel-attention MLP -- forward and backward.  All HBM/latency-bound, NHWC fp32, float4 per thread.
#include "lhn_common.h"

static inline int grid_cap(int64_t blocks, int cap_per_cu) {
  const int64_t cap = (int64_t)lhn_num_cus() * cap_per_cu;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}
static inline bool pow2i(int v) { return v > 0 && (v & (v - 1)) == 0; }

// ------------------------------------------------------------------ BatchNorm finalize
// torch.nn.BatchNorm2d: normalise with the biased batch variance, update running_var with the
// unbiased one, eps inside the sqrt, momentum = exponential average factor.
// One workgroup of up to 1024 threads: thread (c, g) folds replicas g, g + G, ... of channel c (all of its loads in flight
// together), the G partial sums of a channel meet in LDS in a fixed order (run-to-run reproducible), one thread per channel
// does the arithmetic.  (The first version walked one thread per channel through 64 dependent double loads: 6 us per launch,
// 52 launches per forward of the MSRB hourglass.)
#define LHN_FIN_THREADS 1024
__global__ void __launch_bounds__(LHN_FIN_THREADS) k_bn_finalize(const double* __restrict__ stats, const float* __restrict__ gamma,
                              const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
                              int64_t* __restrict__ nbt, float* __restrict__ table, int cs, int coff, int C,
                              float* __restrict__ save, double count, float eps, float momentum, float slope, int training,
                              const float* __restrict__ cbias, int SC) {
  // SC = channels of the statistics / save layout (>= C: a convolution whose output view is padded to a multiple of 4
  // accumulates SC columns, the BatchNorm owns the first C)
  __shared__ double part[2 * LHN_FIN_THREADS];
  const int nt = blockDim.x;
  int G = nt / C;
  G = G < 1 ? 1 : (G > LHN_STAT_REPLICAS ? LHN_STAT_REPLICAS : G);
  for (int c0 = 0; c0 < C; c0 += nt) {           // (one round unless C > 1024)
    const int g = threadIdx.x / C, c = c0 + (G > 1 ? threadIdx.x - g * C : threadIdx.x);
    if (c0) __syncthreads();
    if (training && g < G && c < C) {
      double s1 = 0, s2 = 0;
#pragma unroll 4
      for (int r = g; r < LHN_STAT_REPLICAS; r += G) {
        s1 += stats[(size_t)r * 2 * SC + c];
        s2 += stats[(size_t)r * 2 * SC + SC + c];
      }
      part[2 * threadIdx.x] = s1;
      part[2 * threadIdx.x + 1] = s2;
    }
    __syncthreads();
    if (g != 0 || c >= C) continue;
    double mean, var;
    const double cb = cbias ? (double)cbias[c] : 0.0;   // bias of the conv in front: stored output excludes it
    if (training) {
      double s1 = 0, s2 = 0;
      for (int k = 0; k < G; ++k) {
        s1 += part[2 * (k * C + threadIdx.x)];
        s2 += part[2 * (k * C + threadIdx.x) + 1];
      }
      mean = s1 / count;
      var = s2 / count - mean * mean;
      if (var < 0) var = 0;
      if (rmean) {
        rmean[c] = (float)((1.0 - (double)momentum) * (double)rmean[c] + (double)momentum * (mean + cb));
        const double unb = count > 1 ? var * count / (count - 1.0) : var;
        rvar[c] = (float)((1.0 - (double)momentum) * (double)rvar[c] + (double)momentum * unb);
      }
    } else {
      mean = (double)rmean[c] - cb;
      var = rvar[c];
    }
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float gm = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = gm * invstd;
    table[coff + c] = sc;
    table[cs + coff + c] = b - (float)mean * sc;
    table[2 * cs + coff + c] = slope;
    if (save) {
      save[c] = (float)mean;
      save[SC + c] = invstd;
    }
  }
  if (training && nbt && threadIdx.x == 0) nbt[0] += 1;
}

__global__ void k_table_fill(float* __restrict__ table, int cs, int coff, int C, float sc, float sh, float sl) {
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    table[coff + c] = sc;
    table[cs + coff + c] = sh;
    table[2 * cs + coff + c] = sl;
  }
}

// pending transform of a biased, BN-free (deployed) convolution: v = lrelu_slope(raw + bias)
__global__ void k_table_bias(float* __restrict__ table, int cs, int coff, int C, const float* __restrict__ bias, float sl) {
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    table[coff + c] = 1.f;
    table[cs + coff + c] = bias ? bias[c] : 0.f;
    table[2 * cs + coff + c] = sl;
  }
}

// ------------------------------------------------------------------ elementwise combine
// dst = lrelu_{out_slope}( sum_i value_i(nearest-resampled to dst geometry) ), dst stored plain.
struct EwSrcs {
  lhn_view v[3];
  float coef[3];     // dst = act(sum_i coef[i] * value_i)
  int mode;          // bit 0: PRODUCT of the sources instead of their sum (lite_hrnet.py:105-107: s * interpolate(a));
                     // bit 1: smaller sources are resampled bilinearly with align_corners=True (lite_hrnet.py:272-274)
};
// bilinear taps of destination index d in a source axis of `in` samples (align_corners=True): i0, i1, weight of i1
__device__ __forceinline__ void bil_taps(int d, int in, int out, int& i0, int& i1, float& w1) {
  if (in == out || out == 1) {
    i0 = i1 = (in == out) ? d : 0;
    w1 = 0.f;
    return;
  }
  const float pos = (float)d * ((float)(in - 1) / (float)(out - 1));
  i0 = (int)floorf(pos);
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + 1 < in ? i0 + 1 : in - 1;
  w1 = pos - (float)i0;
}
// activation codes of the combine beyond a leaky slope: LHN_SLOPE_SILU (2), LHN_SLOPE_RELU_SIGMOID (3) = sigmoid(relu(v))
// (lite_hrnet.py:62-70,86-97: nn.ReLU followed by nn.Sigmoid)
__device__ __forceinline__ float lhn_relu_sigmoid(float v) { return 1.f / (1.f + expf(-fmaxf(v, 0.f))); }
__device__ __forceinline__ float lhn_relu_sigmoid_grad(float v) {
  if (!(v > 0.f)) return 0.f;
  const float s = 1.f / (1.f + expf(-v));
  return s * (1.f - s);
}
__device__ __forceinline__ int nearest_src(int d, int in, int out) {
  if (in == out) return d;
  const float sc = (float)in / (float)out;
  const int s = (int)floorf((float)d * sc);
  return s < in - 1 ? s : in - 1;
}
template <bool BIL>
__global__ void __launch_bounds__(256, BIL ? 1 : 3) k_ew_fwd(EwSrcs S, int nsrc, lhn_view dst, float out_slope) {
  const int C4 = dst.C >> 2, c4 = threadIdx.x % C4, pl = threadIdx.x / C4, PL = 256 / C4;
  const int rows = dst.N * dst.H;
  Xf4 xf[3];
  int ca[3];
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (k < nsrc) {
      ca[k] = S.v[k].coff + 4 * c4;
      xf[k] = lhn_load_xf(S.v[k], ca[k]);
    }
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const int n = row / dst.H, h = row - n * dst.H;
    const float* base[3];
    f4 gate[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (k < nsrc) {
        const lhn_view& v = S.v[k];
        const int hs = nearest_src(h, v.H, dst.H);
        base[k] = v.data + ((size_t)(n * v.H + hs) * v.W) * v.cstride + ca[k];
        gate[k] = v.gate ? *reinterpret_cast<const f4*>(v.gate + (size_t)n * v.cstride + ca[k]) : (f4){1.f, 1.f, 1.f, 1.f};
      }
    float* out = dst.data + (size_t)row * dst.W * dst.cstride + dst.coff + 4 * c4;
    const bool mul = (S.mode & 1) != 0;
    constexpr bool bil = BIL;
    if (!bil) {
      // plain / nearest-upsampled operands: four pixels per thread and pass, every load of the batch in flight together
      // (one pixel at a time the 64x64 upsample-add ran at 2.3 TB/s: two dependent loads per 16-pixel step)
      for (int w0 = LHN_LANE0(pl, PL); w0 < dst.W; w0 += 4 * PL) {
        f4 raw[3][4];
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (k < nsrc) {
            const lhn_view& v = S.v[k];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int w = min(w0 + j * PL, dst.W - 1);
              const int ws = (v.W == dst.W) ? w : nearest_src(w, v.W, dst.W);
              raw[k][j] = *reinterpret_cast<const f4*>(base[k] + (size_t)ws * v.cstride);
            }
          }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int w = w0 + j * PL;
          if (w < dst.W) {
            f4 acc = mul ? (f4){1.f, 1.f, 1.f, 1.f} : (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 3; ++k)
              if (k < nsrc) {
                const f4 val = lhn_apply_xf(raw[k][j], xf[k]) * (gate[k] * S.coef[k]);
                if (mul) acc *= val; else acc += val;
              }
            if (out_slope == LHN_SLOPE_RELU_SIGMOID) {
              acc.x = lhn_relu_sigmoid(acc.x); acc.y = lhn_relu_sigmoid(acc.y); acc.z = lhn_relu_sigmoid(acc.z); acc.w = lhn_relu_sigmoid(acc.w);
            } else if (out_slope == LHN_SLOPE_SILU) {
              acc.x = lhn_silu(acc.x); acc.y = lhn_silu(acc.y); acc.z = lhn_silu(acc.z); acc.w = lhn_silu(acc.w);
            } else {
              acc.x = lhn_lrelu(acc.x, out_slope); acc.y = lhn_lrelu(acc.y, out_slope); acc.z = lhn_lrelu(acc.z, out_slope); acc.w = lhn_lrelu(acc.w, out_slope);
            }
            *reinterpret_cast<f4*>(out + (size_t)w * dst.cstride) = acc;
          }
        }
      }
      continue;
    }
    for (int w = LHN_LANE0(pl, PL); w < dst.W; w += PL) {
      f4 acc = mul ? (f4){1.f, 1.f, 1.f, 1.f} : (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (k < nsrc) {
          const lhn_view& v = S.v[k];
          f4 val;
          if (bil && (v.H != dst.H || v.W != dst.W)) {
            int h0, h1, w0, w1;
            float ah, aw;
            bil_taps(h, v.H, dst.H, h0, h1, ah);
            bil_taps(w, v.W, dst.W, w0, w1, aw);
            const float* b0 = v.data + ((size_t)(n * v.H + h0) * v.W) * v.cstride + ca[k];
            const float* b1 = v.data + ((size_t)(n * v.H + h1) * v.W) * v.cstride + ca[k];
            const f4 v00 = lhn_apply_xf(*reinterpret_cast<const f4*>(b0 + (size_t)w0 * v.cstride), xf[k]);
            const f4 v01 = lhn_apply_xf(*reinterpret_cast<const f4*>(b0 + (size_t)w1 * v.cstride), xf[k]);
            const f4 v10 = lhn_apply_xf(*reinterpret_cast<const f4*>(b1 + (size_t)w0 * v.cstride), xf[k]);
            const f4 v11 = lhn_apply_xf(*reinterpret_cast<const f4*>(b1 + (size_t)w1 * v.cstride), xf[k]);
            // torch's upsample_bilinear2d order: rows first, then columns
            val = ((v00 * (1.f - aw) + v01 * aw) * (1.f - ah) + (v10 * (1.f - aw) + v11 * aw) * ah) * (gate[k] * S.coef[k]);
          } else {
            const int ws = nearest_src(w, v.W, dst.W);
            val = lhn_apply_xf(*reinterpret_cast<const f4*>(base[k] + (size_t)ws * v.cstride), xf[k]) * (gate[k] * S.coef[k]);
          }
          if (mul) acc *= val; else acc += val;
        }
      if (out_slope == LHN_SLOPE_RELU_SIGMOID) {
        acc.x = lhn_relu_sigmoid(acc.x);
        acc.y = lhn_relu_sigmoid(acc.y);
        acc.z = lhn_relu_sigmoid(acc.z);
        acc.w = lhn_relu_sigmoid(acc.w);
      } else if (out_slope == LHN_SLOPE_SILU) {
        acc.x = lhn_silu(acc.x);
        acc.y = lhn_silu(acc.y);
        acc.z = lhn_silu(acc.z);
        acc.w = lhn_silu(acc.w);
      } else {
        acc.x = lhn_lrelu(acc.x, out_slope);
        acc.y = lhn_lrelu(acc.y, out_slope);
        acc.z = lhn_lrelu(acc.z, out_slope);
        acc.w = lhn_lrelu(acc.w, out_slope);
      }
      *reinterpret_cast<f4*>(out + (size_t)w * dst.cstride) = acc;
    }
  }
}

// backward of the combine: d(value_i) (+)= d(dst) * lrelu'(dst); upsampled sources sum their fan-out.
// One launch per source (gather form: each source element sums the dst elements that read it).
__global__ void __launch_bounds__(256) k_ew_bwd_src(lhn_view src, lhn_view dst, const float* __restrict__ ddst,
                                                    const float* __restrict__ dst_dpool, float out_slope,
                                                    float* __restrict__ dsrc, int accumulate, lhn_bnsum bs) {
  __shared__ f4 bred[512];
  const int C4 = src.C >> 2, c4 = threadIdx.x % C4, pl = threadIdx.x / C4, PL = 256 / C4;
  // bs.sums: this launch's part of the BatchNorm-backward sums of the convolution that produced src (see lhn_bnsum)
  f4 bsum = (f4){0.f, 0.f, 0.f, 0.f}, bsq = bsum, bmean = bsum, binv = bsum;
  Xf4 bxf;
  if (bs.sums) {
    bxf = lhn_load_xf(src, src.coff + 4 * c4);
    bmean = *reinterpret_cast<const f4*>(bs.save + bs.coff + 4 * c4);
    binv = *reinterpret_cast<const f4*>(bs.save + bs.C + bs.coff + 4 * c4);
  }
  const int fh = dst.H / src.H, fw = dst.W / src.W;  // integer fan-out (host checks divisibility)
  const int cd = dst.coff + 4 * c4;
  const int rows = src.N * src.H;
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const int n = row / src.H, h = row - n * src.H;
    const f4 gate = dst.gate ? *reinterpret_cast<const f4*>(dst.gate + (size_t)n * dst.cstride + cd) : (f4){1.f, 1.f, 1.f, 1.f};
    for (int w = LHN_LANE0(pl, PL); w < src.W; w += PL) {
      f4 g = (f4){0.f, 0.f, 0.f, 0.f};
      for (int a = 0; a < fh; ++a)
        for (int b = 0; b < fw; ++b) {
          const int hd = h * fh + a, wd = w * fw + b;
          const size_t pd = ((size_t)(n * dst.H + hd) * dst.W + wd) * dst.cstride + cd;
          f4 e = *reinterpret_cast<const f4*>(ddst + pd) * gate;
          if (dst_dpool) {
            lhn_gradview gv{nullptr, dst_dpool, nullptr};
            e += lhn_dpool_sum(gv, dst, n, hd, wd, cd);
          }
          if (out_slope == LHN_SLOPE_SILU || out_slope == LHN_SLOPE_RELU_SIGMOID) {
            // single same-size source (host-checked): recompute the pre-activation from it
            const Xf4 sxf = lhn_load_xf(src, src.coff + 4 * c4);
            const f4 v = lhn_load_val(src, sxf, (int64_t)(n * src.H + h) * src.W + w, n, src.coff + 4 * c4);
            if (out_slope == LHN_SLOPE_SILU) {
              e.x *= lhn_silu_grad(v.x);
              e.y *= lhn_silu_grad(v.y);
              e.z *= lhn_silu_grad(v.z);
              e.w *= lhn_silu_grad(v.w);
            } else {
              e.x *= lhn_relu_sigmoid_grad(v.x);
              e.y *= lhn_relu_sigmoid_grad(v.y);
              e.z *= lhn_relu_sigmoid_grad(v.z);
              e.w *= lhn_relu_sigmoid_grad(v.w);
            }
          } else if (out_slope != 1.f) {
            const f4 o = *reinterpret_cast<const f4*>(dst.data + pd);
            e.x *= o.x > 0.f ? 1.f : out_slope;
            e.y *= o.y > 0.f ? 1.f : out_slope;
            e.z *= o.z > 0.f ? 1.f : out_slope;
            e.w *= o.w > 0.f ? 1.f : out_slope;
          }
          g += e;
        }
      float* o = dsrc + ((size_t)row * src.W + w) * src.cstride + src.coff + 4 * c4;
      if (bs.sums) {
        const f4 raw = *reinterpret_cast<const f4*>(src.data + ((size_t)row * src.W + w) * src.cstride + src.coff + 4 * c4);
        const f4 du = g * lhn_dact_xf(raw, bxf);
        bsum += du;
        bsq += du * ((raw - bmean) * binv);
      }
      if (accumulate) g += *reinterpret_cast<const f4*>(o);
      *reinterpret_cast<f4*>(o) = g;
    }
  }
  if (bs.sums) lhn_bns_flush(bs, bsum, bsq, C4, bred);
}

// The same for up to THREE sources of the destination's own resolution in one pass (residual sums: d(dst) * lrelu'(dst) is the
// gradient of every one of them): d(dst) and dst are read once instead of once per source.
struct EwBwdMulti {
  int n;
  lhn_view src[3];
  float* dsrc[3];
  int acc[3];
  lhn_bnsum bs[3];
};
__global__ void __launch_bounds__(256) k_ew_bwd_multi(EwBwdMulti m, lhn_view dst, const float* __restrict__ ddst,
                                                      const float* __restrict__ dst_dpool, float out_slope) {
  __shared__ f4 bred[512];
  const int C4 = dst.C >> 2, c4 = threadIdx.x % C4, pl = threadIdx.x / C4, PL = 256 / C4;
  f4 bsum[3], bsq[3], bmean[3], binv[3];
  Xf4 bxf[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    bsum[k] = bsq[k] = bmean[k] = binv[k] = (f4){0.f, 0.f, 0.f, 0.f};
    if (k < m.n && m.bs[k].sums) {
      bxf[k] = lhn_load_xf(m.src[k], m.src[k].coff + 4 * c4);
      bmean[k] = *reinterpret_cast<const f4*>(m.bs[k].save + m.bs[k].coff + 4 * c4);
      binv[k] = *reinterpret_cast<const f4*>(m.bs[k].save + m.bs[k].C + m.bs[k].coff + 4 * c4);
    }
  }
  const int cd = dst.coff + 4 * c4;
  const int rows = dst.N * dst.H;
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const int n = row / dst.H, h = row - n * dst.H;
    const f4 gate = dst.gate ? *reinterpret_cast<const f4*>(dst.gate + (size_t)n * dst.cstride + cd) : (f4){1.f, 1.f, 1.f, 1.f};
    for (int w = LHN_LANE0(pl, PL); w < dst.W; w += PL) {
      const size_t pix = (size_t)row * dst.W + w, pd = pix * dst.cstride + cd;
      f4 e = *reinterpret_cast<const f4*>(ddst + pd) * gate;
      if (dst_dpool) {
        lhn_gradview gv{nullptr, dst_dpool, nullptr};
        e += lhn_dpool_sum(gv, dst, n, h, w, cd);
      }
      if (out_slope != 1.f) {
        const f4 o = *reinterpret_cast<const f4*>(dst.data + pd);
        e.x *= o.x > 0.f ? 1.f : out_slope;
        e.y *= o.y > 0.f ? 1.f : out_slope;
        e.z *= o.z > 0.f ? 1.f : out_slope;
        e.w *= o.w > 0.f ? 1.f : out_slope;
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (k < m.n) {
          const size_t ps = pix * m.src[k].cstride + m.src[k].coff + 4 * c4;
          f4 g = e;
          if (m.bs[k].sums) {
            const f4 raw = *reinterpret_cast<const f4*>(m.src[k].data + ps);
            const f4 du = g * lhn_dact_xf(raw, bxf[k]);
            bsum[k] += du;
            bsq[k] += du * ((raw - bmean[k]) * binv[k]);
          }
          float* o = m.dsrc[k] + ps;
          if (m.acc[k]) g += *reinterpret_cast<const f4*>(o);
          *reinterpret_cast<f4*>(o) = g;
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (k < m.n && m.bs[k].sums) {
      lhn_bns_flush(m.bs[k], bsum[k], bsq[k], C4, bred);
      __syncthreads();
    }
}

// ------------------------------------------------------------------ 2x2 stride-2 max pool (ceil_mode)
__global__ void __launch_bounds__(256) k_maxpool2_fwd(lhn_view x, lhn_view y) {
  const int C4 = y.C >> 2, c4 = threadIdx.x % C4, pl = threadIdx.x / C4, PL = 256 / C4;
  const int ca = x.coff + 4 * c4;
  const Xf4 xf = lhn_load_xf(x, ca);
  const int rows = y.N * y.H;
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const int n = row / y.H, ho = row - n * y.H;
    const f4 gate = x.gate ? *reinterpret_cast<const f4*>(x.gate + (size_t)n * x.cstride + ca) : (f4){1.f, 1.f, 1.f, 1.f};
    for (int wo = LHN_LANE0(pl, PL); wo < y.W; wo += PL) {
      f4 m = (f4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int ih = 2 * ho + (o >> 1), iw = 2 * wo + (o & 1);
        if (ih < x.H && iw < x.W) {
          const f4 v = lhn_apply_xf(*reinterpret_cast<const f4*>(x.data + ((size_t)(n * x.H + ih) * x.W + iw) * x.cstride + ca), xf) * gate;
          m.x = v.x > m.x || v.x != v.x ? v.x : m.x;
          m.y = v.y > m.y || v.y != v.y ? v.y : m.y;
          m.z = v.z > m.z || v.z != v.z ? v.z : m.z;
          m.w = v.w > m.w || v.w != v.w ? v.w : m.w;
        }
      }
      *reinterpret_cast<f4*>(y.data + ((size_t)row * y.W + wo) * y.cstride + y.coff + 4 * c4) = m;
    }
  }
}
// gradient goes to the first window element (scan order) holding the max.  One thread per OUTPUT element
// re-evaluates its 2x2 window (every input pixel belongs to exactly one window), so the comparison never
// depends on two kernels rounding the pending transform identically.
// ADD: the gradients of other readers of x (lhn_grad_adds) join the same store -- one pass over d(x) instead of three
template <bool ADD>
__global__ void __launch_bounds__(256) k_maxpool2_bwd(lhn_view x, lhn_view y, const float* __restrict__ dy,
                                                      float* __restrict__ dx, int accumulate, lhn_bnsum bs, lhn_grad_adds ad) {
  __shared__ f4 bred[512];
  const int C4 = y.C >> 2, c4 = threadIdx.x % C4, pl = threadIdx.x / C4, PL = 256 / C4;
  const int ca = x.coff + 4 * c4;
  const Xf4 xf = lhn_load_xf(x, ca);
  f4 bsum = (f4){0.f, 0.f, 0.f, 0.f}, bsq = bsum, bmean = bsum, binv = bsum;      // lhn_bnsum: the producer of x
  if (bs.sums) {
    bmean = *reinterpret_cast<const f4*>(bs.save + bs.coff + 4 * c4);
    binv = *reinterpret_cast<const f4*>(bs.save + bs.C + bs.coff + 4 * c4);
  }
  const int rows = y.N * y.H;
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const int n = row / y.H, ho = row - n * y.H;
    const f4 gate = x.gate ? *reinterpret_cast<const f4*>(x.gate + (size_t)n * x.cstride + ca) : (f4){1.f, 1.f, 1.f, 1.f};
    for (int wo = LHN_LANE0(pl, PL); wo < y.W; wo += PL) {
      const f4 g = *reinterpret_cast<const f4*>(dy + ((size_t)row * y.W + wo) * y.cstride + y.coff + 4 * c4);
      f4 m = (f4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      int arg[4] = {-1, -1, -1, -1};
      f4 raws[4];
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int ih = 2 * ho + (o >> 1), iw = 2 * wo + (o & 1);
        if (ih < x.H && iw < x.W) {
          raws[o] = *reinterpret_cast<const f4*>(x.data + ((size_t)(n * x.H + ih) * x.W + iw) * x.cstride + ca);
          const f4 v = lhn_apply_xf(raws[o], xf) * gate;
          if (v.x > m.x || v.x != v.x || arg[0] < 0) { if (!(m.x != m.x)) { m.x = v.x; arg[0] = o; } }
          if (v.y > m.y || v.y != v.y || arg[1] < 0) { if (!(m.y != m.y)) { m.y = v.y; arg[1] = o; } }
          if (v.z > m.z || v.z != v.z || arg[2] < 0) { if (!(m.z != m.z)) { m.z = v.z; arg[2] = o; } }
          if (v.w > m.w || v.w != v.w || arg[3] < 0) { if (!(m.w != m.w)) { m.w = v.w; arg[3] = o; } }
        }
      }
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int ih = 2 * ho + (o >> 1), iw = 2 * wo + (o & 1);
        if (ih < x.H && iw < x.W) {
          f4 r = (f4){arg[0] == o ? g.x : 0.f, arg[1] == o ? g.y : 0.f, arg[2] == o ? g.z : 0.f, arg[3] == o ? g.w : 0.f};
          if (ADD) {
            if (ad.same) r += *reinterpret_cast<const f4*>(ad.same + ((size_t)(n * x.H + ih) * x.W + iw) * ad.same_cstride + ad.same_coff + 4 * c4);
            if (ad.pooled) {      // the bins of lhn_avgpool_bwd that contain (ih, iw)
              const int OH = ad.OH, OW = ad.OW, ohc = (ih * OH) / x.H, owc = (iw * OW) / x.W;
              for (int oh = max(ohc - 1, 0); oh <= min(ohc + 1, OH - 1); ++oh) {
                const int h0 = (oh * x.H) / OH, h1 = ((oh + 1) * x.H + OH - 1) / OH;
                if (ih < h0 || ih >= h1) continue;
                for (int ow = max(owc - 1, 0); ow <= min(owc + 1, OW - 1); ++ow) {
                  const int w0 = (ow * x.W) / OW, w1 = ((ow + 1) * x.W + OW - 1) / OW;
                  if (iw < w0 || iw >= w1) continue;
                  const float inv = 1.f / (float)((h1 - h0) * (w1 - w0));
                  r += *reinterpret_cast<const f4*>(ad.pooled + ((size_t)(n * OH + oh) * OW + ow) * ad.pooled_cstride + ad.pooled_coff + 4 * c4) * inv;
                }
              }
            }
          }
          float* q = dx + ((size_t)(n * x.H + ih) * x.W + iw) * x.cstride + ca;
          if (bs.sums) {
            const f4 du = r * lhn_dact_xf(raws[o], xf);
            bsum += du;
            bsq += du * ((raws[o] - bmean) * binv);
          }
          if (accumulate) r += *reinterpret_cast<const f4*>(q);
          *reinterpret_cast<f4*>(q) = r;
        }
      }
    }
  }
  if (bs.sums) lhn_bns_flush(bs, bsum, bsq, C4, bred);
}

// BatchNorms behind channel slices of one buffer (a gated buffer is written by up to two convolutions, each with its own
// BatchNorm): saved (mean | invstd) per slice, optionally the slice's BatchNorm-backward sums.  Channels outside every slice
// (a pass-through half) read mean 0, invstd 1.
struct BnSlices {
  const float* save[2];   // [2][C[k]]
  double* sums[2];        // [LHN_STAT_REPLICAS][2][C[k]] or NULL
  int lo[2], C[2], n;
};
__device__ __forceinline__ void lhn_slice_mi4(const BnSlices& sl, int c, f4& mean, f4& inv) {
  mean = (f4){0.f, 0.f, 0.f, 0.f};
  inv = (f4){1.f, 1.f, 1.f, 1.f};
#pragma unroll
  for (int k = 0; k < 2; ++k)
    if (k < sl.n && c >= sl.lo[k] && c < sl.lo[k] + sl.C[k]) {
      mean = *reinterpret_cast<const f4*>(sl.save[k] + (c - sl.lo[k]));
      inv = *reinterpret_cast<const f4*>(sl.save[k] + sl.C[k] + (c - sl.lo[k]));
    }
}
__device__ __forceinline__ f4 lhn_dact4(f4 raw, const Xf4& t) {      // derivative of the pending activation at raw
  const f4 u = raw * t.sc + t.sh;
  return (f4){u.x > 0.f ? 1.f : t.sl.x, u.y > 0.f ? 1.f : t.sl.y, u.z > 0.f ? 1.f : t.sl.z, u.w > 0.f ? 1.f : t.sl.w};
}

// ------------------------------------------------------------------ adaptive average pool -> dense [N,OH,OW,C]
// STAT (channel attention of a training plan): besides the bin means of the consumed value the kernel writes, per (n, bin, c),
//   M0 = sum over the bin of act'(u)     and     M1 = sum over the bin of act'(u) * xhat        (pstat[N*OH*OW][2][C])
// -- with them the BatchNorm-backward sums of the gated buffer follow from the gate-gradient pass alone (k_ca_bwd1), no
// second pass over the feature map and its gradient.
// COPY (gated RepBasicUnit, litehourglass.py:74-77 `ca(cat(left, right))`): channels [0, src.C) of the pooled buffer are the
// pass-through half, which nobody has written yet -- this kernel reads them from `src` (pending transform and gate applied),
// pools them AND stores them into x (each pixel by the one bin that owns it): the separate copy pass and its re-read disappear.
template <bool STAT, bool COPY>
__global__ void __launch_bounds__(256) k_avgpool_fwd(lhn_view x, float* __restrict__ out, int OH, int OW, int ostride,
                                                     int ocoff, float* __restrict__ pstat, BnSlices sl, lhn_view src) {
  __shared__ f4 red[STAT ? 768 : 256];
  const int C4 = x.C >> 2, PL = 256 / C4;
  const int b = blockIdx.x;
  const int ow = b % OW, oh = (b / OW) % OH, n = b / (OW * OH);
  const int h0 = (oh * x.H) / OH, h1 = ((oh + 1) * x.H + OH - 1) / OH;
  const int w0 = (ow * x.W) / OW, w1 = ((ow + 1) * x.W + OW - 1) / OW;
  const int bw = w1 - w0, cnt = (h1 - h0) * bw;
  const int c4 = threadIdx.x % C4, pl = threadIdx.x / C4;
  const int ca = x.coff + 4 * c4;
  const bool cp = COPY && 4 * c4 < src.C;                    // this thread's channels come from src
  const int sca = src.coff + 4 * c4;
  Xf4 xf = lhn_load_xf(x, ca);
  f4 sgate = (f4){1.f, 1.f, 1.f, 1.f};
  if (cp) {
    xf = lhn_load_xf(src, sca);
    if (src.gate) sgate = *reinterpret_cast<const f4*>(src.gate + (int64_t)n * src.cstride + sca);
  }
  f4 mean, inv;
  if (STAT) lhn_slice_mi4(sl, ca, mean, inv);
  // four independent loads in flight per thread; the (per-image) gate factors out of the sum
  const float* base = cp ? src.data + (int64_t)n * x.H * x.W * src.cstride + sca : x.data + (int64_t)n * x.H * x.W * x.cstride + ca;
  const int64_t pstr = cp ? src.cstride : x.cstride;
  auto at = [&](int p) { return *reinterpret_cast<const f4*>(base + (int64_t)((h0 + p / bw) * x.W + w0 + p % bw) * pstr); };
  // a pixel is stored by the bin whose exclusive range [lo(oh), lo(oh + 1)) x [lo(ow), lo(ow + 1)) holds it (bins overlap)
  const int hown = ((oh + 1) * x.H) / OH, wown = ((ow + 1) * x.W) / OW;
  float* xout = x.data + (int64_t)n * x.H * x.W * x.cstride + ca;
  auto put = [&](int p, f4 v) {
    const int h = h0 + p / bw, w = w0 + p % bw;
    if (h < hown && w < wown) *reinterpret_cast<f4*>(xout + (int64_t)(h * x.W + w) * x.cstride) = v;
  };
  f4 s0 = (f4){0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0, m0 = s0, m1 = s0;
  int p = LHN_LANE0(pl, PL);
  for (; p + 3 * PL < cnt; p += 4 * PL) {
    const f4 a = at(p), b2 = at(p + PL), c2 = at(p + 2 * PL), d2 = at(p + 3 * PL);
    const f4 va = lhn_apply_xf(a, xf) * sgate, vb = lhn_apply_xf(b2, xf) * sgate, vc = lhn_apply_xf(c2, xf) * sgate, vd = lhn_apply_xf(d2, xf) * sgate;
    s0 += va;
    s1 += vb;
    s2 += vc;
    s3 += vd;
    if (cp) {
      put(p, va);
      put(p + PL, vb);
      put(p + 2 * PL, vc);
      put(p + 3 * PL, vd);
    }
    if (STAT) {
      const f4 da = lhn_dact4(a, xf), db = lhn_dact4(b2, xf), dc = lhn_dact4(c2, xf), dd = lhn_dact4(d2, xf);
      m0 += (da + db) + (dc + dd);
      m1 += (da * ((a - mean) * inv) + db * ((b2 - mean) * inv)) + (dc * ((c2 - mean) * inv) + dd * ((d2 - mean) * inv));
    }
  }
  for (; p < cnt; p += PL) {
    const f4 a = at(p);
    const f4 va = lhn_apply_xf(a, xf) * sgate;
    s0 += va;
    if (cp) put(p, va);
    if (STAT) {
      const f4 da = lhn_dact4(a, xf);
      m0 += da;
      m1 += da * ((a - mean) * inv);
    }
  }
  f4 s = (s0 + s1) + (s2 + s3);
  if (x.gate) s *= *reinterpret_cast<const f4*>(x.gate + (int64_t)n * x.cstride + ca);
  // power-of-two C4 <= 64: lanes of a wave that share c4 meet by xor-shuffles, the four waves through LDS; any other C4
  // (C4 = 5, 10, 20, 40, 80 ...): one LDS slot per thread
  const bool shuf = C4 <= 64 && (C4 & (C4 - 1)) == 0;
  if (shuf)
    for (int o = C4; o < 64; o <<= 1) {
      s.x += __shfl_xor(s.x, o, 64); s.y += __shfl_xor(s.y, o, 64); s.z += __shfl_xor(s.z, o, 64); s.w += __shfl_xor(s.w, o, 64);
      if (STAT) {
        m0.x += __shfl_xor(m0.x, o, 64); m0.y += __shfl_xor(m0.y, o, 64); m0.z += __shfl_xor(m0.z, o, 64); m0.w += __shfl_xor(m0.w, o, 64);
        m1.x += __shfl_xor(m1.x, o, 64); m1.y += __shfl_xor(m1.y, o, 64); m1.z += __shfl_xor(m1.z, o, 64); m1.w += __shfl_xor(m1.w, o, 64);
      }
    }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (shuf && lane < C4) {
    red[wave * C4 + lane] = s;
    if (STAT) {
      red[256 + wave * C4 + lane] = m0;
      red[512 + wave * C4 + lane] = m1;
    }
  }
  if (!shuf) {
    red[threadIdx.x] = s;
    if (STAT) {
      red[256 + threadIdx.x] = m0;
      red[512 + threadIdx.x] = m1;
    }
  }
  __syncthreads();
  if (threadIdx.x < C4) {
    const int nj = shuf ? 4 : PL;
    f4 t = (f4){0.f, 0.f, 0.f, 0.f}, t0 = t, t1 = t;
    for (int j = 0; j < nj; ++j) {
      t += red[j * C4 + threadIdx.x];
      if (STAT) {
        t0 += red[256 + j * C4 + threadIdx.x];
        t1 += red[512 + j * C4 + threadIdx.x];
      }
    }
    const float invc = 1.f / (float)cnt;
    *reinterpret_cast<f4*>(out + (int64_t)b * ostride + ocoff + 4 * threadIdx.x) = t * invc;
    if (STAT) {
      *reinterpret_cast<f4*>(pstat + ((int64_t)b * 2 + 0) * x.C + 4 * threadIdx.x) = t0;
      *reinterpret_cast<f4*>(pstat + ((int64_t)b * 2 + 1) * x.C + 4 * threadIdx.x) = t1;
    }
  }
}
// Small bins (<= 16 pixels: the 2x2 / 4x4 bins of lite_hrnet.py:56-60, every branch pooled to the smallest map): one THREAD
// per (bin, 4 channels) instead of one workgroup per bin -- 65,536 workgroups of 4 pixels each took 121 us for a 42 MB map.
__global__ void __launch_bounds__(256) k_avgpool_small(lhn_view x, float* __restrict__ out, int OH, int OW, int ostride, int ocoff) {
  const int C4 = x.C >> 2;
  const int64_t total = (int64_t)x.N * OH * OW * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    const int64_t b = i / C4;
    const int ow = (int)(b % OW), oh = (int)((b / OW) % OH), n = (int)(b / ((int64_t)OW * OH));
    const int h0 = (oh * x.H) / OH, h1 = ((oh + 1) * x.H + OH - 1) / OH;
    const int w0 = (ow * x.W) / OW, w1 = ((ow + 1) * x.W + OW - 1) / OW;
    const int ca = x.coff + 4 * c4;
    const Xf4 xf = lhn_load_xf(x, ca);
    f4 s = (f4){0.f, 0.f, 0.f, 0.f};
    for (int h = h0; h < h1; ++h)
      for (int w = w0; w < w1; ++w)
        s += lhn_apply_xf(*reinterpret_cast<const f4*>(x.data + ((int64_t)(n * x.H + h) * x.W + w) * x.cstride + ca), xf);
    if (x.gate) s *= *reinterpret_cast<const f4*>(x.gate + (int64_t)n * x.cstride + ca);
    *reinterpret_cast<f4*>(out + b * ostride + ocoff + 4 * c4) = s * (1.f / (float)((h1 - h0) * (w1 - w0)));
  }
}
// d(value of x) (+)= sum over bins containing the pixel of dout[bin]/|bin|
__global__ void __launch_bounds__(256) k_avgpool_bwd(lhn_view x, const float* __restrict__ dout, int OH, int OW,
                                                     float* __restrict__ dx, int accumulate, int ostride, int ocoff, lhn_bnsum bs) {
  __shared__ f4 bred[512];
  const int C4 = x.C >> 2, c4 = threadIdx.x % C4, pl = threadIdx.x / C4, PL = 256 / C4;
  f4 bsum = (f4){0.f, 0.f, 0.f, 0.f}, bsq = bsum, bmean = bsum, binv = bsum;      // lhn_bnsum: the producer of x
  Xf4 bxf;
  if (bs.sums) {
    bxf = lhn_load_xf(x, x.coff + 4 * c4);
    bmean = *reinterpret_cast<const f4*>(bs.save + bs.coff + 4 * c4);
    binv = *reinterpret_cast<const f4*>(bs.save + bs.C + bs.coff + 4 * c4);
  }
  const int rows = x.N * x.H;
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const int n = row / x.H, h = row - n * x.H;
    for (int w = LHN_LANE0(pl, PL); w < x.W; w += PL) {
      f4 g = (f4){0.f, 0.f, 0.f, 0.f};
      // bin oh contains h iff floor(oh*H/OH) <= h < ceil((oh+1)*H/OH): candidates are floor(h*OH/H) and its neighbours
      const int ohc = (h * OH) / x.H, owc = (w * OW) / x.W;
      for (int oh = max(ohc - 1, 0); oh <= min(ohc + 1, OH - 1); ++oh) {
        const int h0 = (oh * x.H) / OH, h1 = ((oh + 1) * x.H + OH - 1) / OH;
        if (h < h0 || h >= h1) continue;
        for (int ow = max(owc - 1, 0); ow <= min(owc + 1, OW - 1); ++ow) {
          const int w0 = (ow * x.W) / OW, w1 = ((ow + 1) * x.W + OW - 1) / OW;
          if (w < w0 || w >= w1) continue;
          const float inv = 1.f / (float)((h1 - h0) * (w1 - w0));
          g += *reinterpret_cast<const f4*>(dout + ((size_t)(n * OH + oh) * OW + ow) * ostride + ocoff + 4 * c4) * inv;
        }
      }
      float* o = dx + ((size_t)row * x.W + w) * x.cstride + x.coff + 4 * c4;
      if (bs.sums) {
        const f4 raw = *reinterpret_cast<const f4*>(x.data + ((size_t)row * x.W + w) * x.cstride + x.coff + 4 * c4);
        const f4 du = g * lhn_dact_xf(raw, bxf);
        bsum += du;
        bsq += du * ((raw - bmean) * binv);
      }
      if (accumulate) g += *reinterpret_cast<const f4*>(o);
      *reinterpret_cast<f4*>(o) = g;
    }
  }
  if (bs.sums) lhn_bns_flush(bs, bsum, bsq, C4, bred);
}

// ------------------------------------------------------------------ channel attention MLP (common.py:40-66)
// save layout (floats): a[N*C] | ahat[N*C] | h[N*C/2] | g[N*C] | mean[C] | invstd[C]
// grid = C/32 blocks; thread = (channel lane 0..31, sample lane 0..NL-1), NL = blockDim.x / 32 (32 as launched: the per-channel
// chains over the batch are latency-bound, 8 lanes took 13.2 us where 32 take 8.4 us)
__global__ void __launch_bounds__(1024) k_ca1(const float* __restrict__ pooled, const float* __restrict__ w3,
                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                             float* __restrict__ rmean, float* __restrict__ rvar, int64_t* __restrict__ nbt,
                                             const float* __restrict__ mask, float* __restrict__ save, int N, int C,
                                             float eps, float momentum, int training, int stage, double* __restrict__ gsum,
                                             double count_scale) {
  // stage 0: everything; SyncBatchNorm splits the kernel around an all-reduce of gsum[2][C] (stage 1: conv + local sums,
  // stage 2: statistics over N*count_scale samples + normalisation)
  __shared__ double rs[32][32], rq[32][32];
  const int NL = blockDim.x >> 5;   // sample lanes (32 at the 1024-thread launch)
  __shared__ float s_sc[32], s_sh[32];
  float* a = save;
  float* ahat = save + (int64_t)N * C;
  float* smean = save + (int64_t)N * C * 3 + (int64_t)N * (C / 2);
  float* sinv = smean + C;
  const int cl = threadIdx.x & 31, nl = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
  const bool ok = c < C;
  float wt[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wt[t] = ok ? w3[c * 9 + t] : 0.f;
  double s = 0, q = 0;
  if (ok && stage != 2)
    for (int n = nl; n < N; n += NL) {
      float v = 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t) v += pooled[((int64_t)n * 9 + t) * C + c] * wt[t];
      a[(int64_t)n * C + c] = v;
      s += v;
      q += (double)v * v;
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
  const double NT = (double)N * count_scale;
  if (nl == 0 && ok) {
    double mean, var;
    if (!gamma) {          // deployed attention (common.py:68-90): the BatchNorm is folded, `beta` is the conv bias
      smean[c] = 0.f;
      sinv[c] = 1.f;
      s_sc[cl] = 1.f;
      s_sh[cl] = beta ? beta[c] : 0.f;
    } else {
    if (training) {
      mean = s / NT;
      var = q / NT - mean * mean;
      if (var < 0) var = 0;
      rmean[c] = (float)((1.0 - (double)momentum) * rmean[c] + (double)momentum * mean);
      rvar[c] = (float)((1.0 - (double)momentum) * rvar[c] + (double)momentum * (NT > 1 ? var * NT / (NT - 1.0) : var));
    } else {
      mean = rmean[c];
      var = rvar[c];
    }
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    smean[c] = (float)mean;
    sinv[c] = invstd;
    s_sc[cl] = gamma[c] * invstd;
    s_sh[cl] = beta[c] - (float)mean * (gamma[c] * invstd);
    }
  }
  __syncthreads();
  if (ok) {
    const float sc = s_sc[cl], sh = s_sh[cl];
    for (int n = nl; n < N; n += NL) {
      float v = a[(int64_t)n * C + c] * sc + sh;
      if (mask) v *= mask[(int64_t)n * C + c];
      ahat[(int64_t)n * C + c] = v;
    }
  }
  if (training && nbt && blockIdx.x == 0 && threadIdx.x == 0) nbt[0] += 1;
}
__global__ void __launch_bounds__(256) k_ca2(const float* __restrict__ w1, const float* __restrict__ b1,
                                             const float* __restrict__ w2, const float* __restrict__ b2,
                                             float* __restrict__ save, float* __restrict__ gate, int gs, int gcoff, int N,
                                             int C) {
  __shared__ float sa[256], shh[128];
  const int n = blockIdx.x, Ch = C / 2;
  const float* ahat = save + (int64_t)N * C + (int64_t)n * C;
  float* h = save + (int64_t)N * C * 2 + (int64_t)n * Ch;
  float* g = save + (int64_t)N * C * 2 + (int64_t)N * Ch + (int64_t)n * C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) sa[c] = ahat[c];
  __syncthreads();
  for (int j = threadIdx.x; j < Ch; j += blockDim.x) {
    float v = b1[j];
    for (int c = 0; c < C; ++c) v += w1[j * C + c] * sa[c];
    v = lhn_lrelu(v, 0.01f);
    h[j] = v;
    shh[j] = v;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float v = b2[c];
    for (int j = 0; j < Ch; ++j) v += w2[c * Ch + j] * shh[j];
    const float sg = 1.f / (1.f + expf(-v));
    g[c] = sg;
    gate[(int64_t)n * gs + gcoff + c] = sg;
  }
}

// d(gate)[n][c] = sum_pixels dz * value_pre_gate   (one block per image)
// tsum != NULL: also T0[n][c] = sum dz * act'(u), T1[n][c] = sum dz * act'(u) * xhat  (tsum[N][2][C]) -- the per-sample halves of
// the gated buffer's BatchNorm-backward sums (see k_avgpool_fwd<STAT> / k_ca_bwd1)
__global__ void __launch_bounds__(256) k_gate_bwd_reduce(lhn_view y, const float* __restrict__ dz,
                                                         float* __restrict__ dgate, float* __restrict__ tsum, BnSlices sl) {
  __shared__ f4 red[768];
  const int C4 = y.C >> 2, PL = 256 / C4, n = blockIdx.x;
  const int c4 = threadIdx.x % C4, pl = threadIdx.x / C4;
  const int ca = y.coff + 4 * c4;
  const Xf4 xf = lhn_load_xf(y, ca);
  f4 mean, inv;
  lhn_slice_mi4(sl, ca, mean, inv);
  const int HW = y.H * y.W;
  const int chunk = (HW + gridDim.y - 1) / gridDim.y;
  const int p0 = blockIdx.y * chunk, p1 = min(HW, p0 + chunk);
  f4 s = (f4){0.f, 0.f, 0.f, 0.f}, t0 = s, t1 = s;
  // two pixels per thread and pass: four independent loads in flight
  for (int p = p0 + LHN_LANE0(pl, PL); p < p1; p += 2 * PL) {
    const int64_t pixa = (int64_t)n * HW + p, pixb = (int64_t)n * HW + min(p + PL, p1 - 1);
    const f4 rawa = *reinterpret_cast<const f4*>(y.data + pixa * y.cstride + ca);
    const f4 rawb = *reinterpret_cast<const f4*>(y.data + pixb * y.cstride + ca);
    const f4 dza = *reinterpret_cast<const f4*>(dz + pixa * y.cstride + ca);
    f4 dzb = *reinterpret_cast<const f4*>(dz + pixb * y.cstride + ca);
    if (p + PL >= p1) dzb = (f4){0.f, 0.f, 0.f, 0.f};
    s += lhn_apply_xf(rawa, xf) * dza + lhn_apply_xf(rawb, xf) * dzb;
    if (tsum) {
      const f4 da = lhn_dact4(rawa, xf) * dza, db = lhn_dact4(rawb, xf) * dzb;
      t0 += da + db;
      t1 += da * ((rawa - mean) * inv) + db * ((rawb - mean) * inv);
    }
  }
  red[threadIdx.x] = s;
  red[256 + threadIdx.x] = t0;
  red[512 + threadIdx.x] = t1;
  __syncthreads();
  if (threadIdx.x < C4) {
    f4 t = (f4){0.f, 0.f, 0.f, 0.f}, u0 = t, u1 = t;
    for (int j = 0; j < PL; ++j) {
      t += red[j * C4 + threadIdx.x];
      u0 += red[256 + j * C4 + threadIdx.x];
      u1 += red[512 + j * C4 + threadIdx.x];
    }
    float* o = dgate + (int64_t)n * y.C + 4 * threadIdx.x;
    atomicAdd(o + 0, t.x);
    atomicAdd(o + 1, t.y);
    atomicAdd(o + 2, t.z);
    atomicAdd(o + 3, t.w);
    if (tsum) {
      float* o0 = tsum + ((int64_t)n * 2 + 0) * y.C + 4 * threadIdx.x;
      float* o1 = tsum + ((int64_t)n * 2 + 1) * y.C + 4 * threadIdx.x;
      atomicAdd(o0 + 0, u0.x); atomicAdd(o0 + 1, u0.y); atomicAdd(o0 + 2, u0.z); atomicAdd(o0 + 3, u0.w);
      atomicAdd(o1 + 0, u1.x); atomicAdd(o1 + 1, u1.y); atomicAdd(o1 + 2, u1.z); atomicAdd(o1 + 3, u1.w);
    }
  }
}

// backward of the MLP; writes dpool[n][bin][cs] (already divided by the bin size) and parameter grads
__global__ void __launch_bounds__(256) k_ca_bwd2(const float* __restrict__ w1, const float* __restrict__ w2,
                                                 const float* __restrict__ save, const float* __restrict__ dgate,
                                                 float* __restrict__ dahat /*[N][C]*/, float* __restrict__ dvbuf /*[N][C]*/,
                                                 float* __restrict__ dhbuf /*[N][C/2]*/, int N, int C) {
  // One workgroup per sample: back through the sigmoid, the second 1x1, the leaky ReLU and the first 1x1.  The parameter
  // gradients (sums over the batch of outer products) are formed by k_ca_bwd1 from the dv / dh rows stored here -- the first
  // version added them with one float atomic per (sample, weight): a million atomics onto 16k addresses per attention.
  __shared__ float sdv[256], sdh[128], shh[128];
  const int Ch = C / 2, n = blockIdx.x;
  const float* h = save + (int64_t)N * C * 2 + (int64_t)n * Ch;
  const float* g = save + (int64_t)N * C * 2 + (int64_t)N * Ch + (int64_t)n * C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float gg = g[c];
    const float dv = dgate[(int64_t)n * C + c] * gg * (1.f - gg);
    sdv[c] = dv;
    dvbuf[(int64_t)n * C + c] = dv;
  }
  for (int j = threadIdx.x; j < Ch; j += blockDim.x) shh[j] = h[j];
  __syncthreads();
  for (int j = threadIdx.x; j < Ch; j += blockDim.x) {
    float d = 0.f;
#pragma unroll 8
    for (int c = 0; c < C; ++c) d += w2[c * Ch + j] * sdv[c];      // (unrolled: the loads of a mat-vec row in flight together)
    d *= shh[j] > 0.f ? 1.f : 0.01f;
    sdh[j] = d;
    dhbuf[(int64_t)n * Ch + j] = d;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float d = 0.f;
#pragma unroll 8
    for (int j = 0; j < Ch; ++j) d += w1[j * C + c] * sdh[j];
    dahat[(int64_t)n * C + c] = d;
  }
}
// grid = C/32 blocks; thread = (channel lane, sample lane) as in k_ca1
__global__ void __launch_bounds__(1024) k_ca_bwd1(const float* __restrict__ pooled, const float* __restrict__ w3,
                                                 const float* __restrict__ gamma, const float* __restrict__ mask,
                                                 const float* __restrict__ save, const float* __restrict__ dahat,
                                                 float* __restrict__ dpool, int cs, int coff, int H, int W,
                                                 float* __restrict__ dw3, float* __restrict__ dgamma,
                                                 float* __restrict__ dbeta, int N, int C, int training, int stage,
                                                 double* __restrict__ gsum, double count_scale, float pgrad_scale,
                                                 const float* __restrict__ tsum, const float* __restrict__ pstat, BnSlices sl,
                                                 const float* __restrict__ dvbuf, const float* __restrict__ dhbuf,
                                                 float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dw2,
                                                 float* __restrict__ db2) {
  // stage 0: everything; SyncBatchNorm: stage 1 = local (sum d, sum d*xhat) -> gsum[2][C], all-reduce, stage 2 = the rest
  const int nmain = (C + 31) / 32;
  if ((int)blockIdx.x >= nmain) {
    // ---- extra workgroups, beside the ones below: parameter gradients of the two 1x1 layers = sums over the batch of the rows
    // k_ca_bwd2 stored.  One thread per weight, samples added in order: one writer per element, no atomics, deterministic.
    if (stage == 1) return;
    const int Ch = C / 2, o = ((int)blockIdx.x - nmain) * blockDim.x + threadIdx.x, nw = C * Ch;
    const float* ahat = save + (int64_t)N * C;
    const float* hh = save + (int64_t)N * C * 2;
    if (o < nw) {                                                         // dW2[c][j] = sum_n dv[n][c] * h[n][j]
      const int cc = o / Ch, j = o - cc * Ch;
      float v = 0.f;
#pragma unroll 8
      for (int n = 0; n < N; ++n) v += dvbuf[(int64_t)n * C + cc] * hh[(int64_t)n * Ch + j];
      dw2[o] += v;
    } else if (o < 2 * nw) {                                              // dW1[j][c] = sum_n dh[n][j] * ahat[n][c]
      const int q = o - nw, j = q / C, cc = q - j * C;
      float v = 0.f;
#pragma unroll 8
      for (int n = 0; n < N; ++n) v += dhbuf[(int64_t)n * Ch + j] * ahat[(int64_t)n * C + cc];
      dw1[q] += v;
    } else if (o < 2 * nw + C) {                                          // db2[c] = sum_n dv[n][c]
      const int cc = o - 2 * nw;
      float v = 0.f;
#pragma unroll 8
      for (int n = 0; n < N; ++n) v += dvbuf[(int64_t)n * C + cc];
      db2[cc] += v;
    } else if (o < 2 * nw + C + Ch) {                                     // db1[j] = sum_n dh[n][j]
      const int j = o - 2 * nw - C;
      float v = 0.f;
#pragma unroll 8
      for (int n = 0; n < N; ++n) v += dhbuf[(int64_t)n * Ch + j];
      db1[j] += v;
    }
    return;
  }
  __shared__ double rs[32][32], rq[32][32];
  __shared__ float rw[32][32][9];
  const int NL = blockDim.x >> 5;   // sample lanes (32 at the 1024-thread launch)
  const float* a = save;
  const float* smean = save + (int64_t)N * C * 3 + (int64_t)N * (C / 2);
  const float* sinv = smean + C;
  const int cl = threadIdx.x & 31, nl = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
  const bool ok = c < C;
  const float mean = ok ? smean[c] : 0.f, invstd = ok ? sinv[c] : 0.f, gm = ok ? gamma[c] : 0.f;
  double sd = 0, sdx = 0;
  if (ok && stage != 2)
    for (int n = nl; n < N; n += NL) {
      float d = dahat[(int64_t)n * C + c];
      if (mask) d *= mask[(int64_t)n * C + c];
      const float xh = (a[(int64_t)n * C + c] - mean) * invstd;
      sd += d;
      sdx += (double)d * xh;
    }
  rs[nl][cl] = sd;
  rq[nl][cl] = sdx;
  __syncthreads();
  sd = 0;
  sdx = 0;
  for (int j = 0; j < NL; ++j) {
    sd += rs[j][cl];
    sdx += rq[j][cl];
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
  const double NT = (double)N * count_scale;
  if (nl == 0 && ok) {
    dgamma[c] += (float)sdx * pgrad_scale;
    dbeta[c] += (float)sd * pgrad_scale;
  }
  // tsum / pstat: the gated buffer's BatchNorm-backward sums, assembled here per channel from the gate-gradient pass (T0, T1),
  // the forward pooling pass (M0, M1 per bin) and the pooled gradient e[bin] = d loss / d pooled[bin] / |bin| formed below:
  //   sum du        = sum_n ( g * T0 + sum_bins e * M0 ),      sum du * xhat = sum_n ( g * T1 + sum_bins e * M1 )
  const float* gsave = save + (int64_t)N * C * 2 + (int64_t)N * (C / 2);      // sigmoid gate g[n][c] of the forward
  double bs0 = 0, bs1 = 0;
  float wt[9], dwt[9], binv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    wt[t] = ok ? w3[c * 9 + t] : 0.f;
    dwt[t] = 0.f;
    const int bi = t / 3, bj = t % 3;
    binv[t] = 1.f / (float)((lhn_bin_hi(bi, H) - lhn_bin_lo(bi, H)) * (lhn_bin_hi(bj, W) - lhn_bin_lo(bj, W)));
  }
  if (ok)
    for (int n = nl; n < N; n += NL) {
      float d = dahat[(int64_t)n * C + c];
      if (mask) d *= mask[(int64_t)n * C + c];
      const float xh = (a[(int64_t)n * C + c] - mean) * invstd;
      float da = gm * invstd * d;
      if (training) da = gm * invstd * (d - (float)(sd / NT) - xh * (float)(sdx / NT));
      float dseg[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        dwt[t] += da * pooled[((int64_t)n * 9 + t) * C + c];
        dseg[t] = da * wt[t] * binv[t];
      }
      lhn_dpool_store(dpool, n, cs, coff + c, dseg);
      if (tsum) {
        const float g = gsave[(int64_t)n * C + c];
        float a0 = g * tsum[((int64_t)n * 2 + 0) * C + c], a1 = g * tsum[((int64_t)n * 2 + 1) * C + c];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          a0 += dseg[t] * pstat[(((int64_t)n * 9 + t) * 2 + 0) * C + c];
          a1 += dseg[t] * pstat[(((int64_t)n * 9 + t) * 2 + 1) * C + c];
        }
        bs0 += a0;
        bs1 += a1;
      }
    }
  if (tsum) {
    __syncthreads();          // rs / rq are free again
    rs[nl][cl] = bs0;
    rq[nl][cl] = bs1;
    __syncthreads();
    if (nl == 0 && ok) {
      double t0 = 0, t1 = 0;
      for (int j = 0; j < NL; ++j) {
        t0 += rs[j][cl];
        t1 += rq[j][cl];
      }
      const int cb = coff + c;                    // channel of the gated buffer
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (k < sl.n && sl.sums[k] && cb >= sl.lo[k] && cb < sl.lo[k] + sl.C[k]) {   // this block is the channel's only writer
          sl.sums[k][cb - sl.lo[k]] = t0;
          sl.sums[k][sl.C[k] + cb - sl.lo[k]] = t1;
        }
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) rw[nl][cl][t] = dwt[t];
  __syncthreads();
  if (nl == 0 && ok) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float v = 0.f;
      for (int j = 0; j < NL; ++j) v += rw[j][cl][t];
      dw3[c * 9 + t] += v;
    }
  }
}

// ------------------------------------------------------------------ BatchNorm backward reductions
// sums[0][c] = sum du, sums[1][c] = sum du * xhat   over all pixels (du: gradient at the BN output)
__global__ void __launch_bounds__(256) k_bn_bwd_reduce(lhn_view y, lhn_gradview g, const float* __restrict__ save,
                                                       double* __restrict__ sums, lhn_bnbwdfin fin) {
  __shared__ f4 red[512];
  const int C4 = y.C >> 2, PL = 256 / C4;
  const int c4 = threadIdx.x % C4, pl = threadIdx.x / C4;
  const int ca = y.coff + 4 * c4;
  const Xf4 xf = lhn_load_xf(y, ca);
  const f4 mean = *reinterpret_cast<const f4*>(save + 4 * c4);
  const f4 inv = *reinterpret_cast<const f4*>(save + y.C + 4 * c4);
  const int rows = y.N * y.H;
  f4 s = (f4){0.f, 0.f, 0.f, 0.f}, q = s;
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const int n = row / y.H, h = row - n * y.H;
    const f4 gate = y.gate ? *reinterpret_cast<const f4*>(y.gate + (size_t)n * y.cstride + ca) : (f4){1.f, 1.f, 1.f, 1.f};
    const size_t rbase = (size_t)row * y.W * y.cstride + ca;
    // four pixels per thread and pass: eight independent loads in flight (one pixel at a time the 64x64 maps ran at 2.7 TB/s)
    for (int w0 = LHN_LANE0(pl, PL); w0 < y.W; w0 += 4 * PL) {
      f4 raw4[4], dz4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const size_t off = rbase + (size_t)min(w0 + j * PL, y.W - 1) * y.cstride;
        raw4[j] = *reinterpret_cast<const f4*>(y.data + off);
        dz4[j] = *reinterpret_cast<const f4*>(g.dz + off);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int w = w0 + j * PL;
        if (w < y.W) {
          const f4 raw = raw4[j];
          f4 e = dz4[j] * gate;
          if (g.dpool) e += lhn_dpool_sum(g, y, n, h, w, ca);
          const f4 u = raw * xf.sc + xf.sh;
          const f4 du = e * (f4){u.x > 0.f ? 1.f : xf.sl.x, u.y > 0.f ? 1.f : xf.sl.y, u.z > 0.f ? 1.f : xf.sl.z, u.w > 0.f ? 1.f : xf.sl.w};
          s += du;
          q += du * ((raw - mean) * inv);
        }
      }
    }
  }
  double* st = sums + (size_t)(blockIdx.x % LHN_STAT_REPLICAS) * 2 * y.C;
  if (C4 <= 32 && (C4 & (C4 - 1)) == 0) {
    lhn_block_stat_atomics(s, q, C4, red, st, st + y.C);
  } else {
    red[threadIdx.x * 2] = s;
    red[threadIdx.x * 2 + 1] = q;
    __syncthreads();
    if (threadIdx.x < C4) {
      double sd[4] = {0, 0, 0, 0}, qd[4] = {0, 0, 0, 0};
      for (int j = 0; j < PL; ++j) {
        const f4 a = red[(j * C4 + threadIdx.x) * 2], b = red[(j * C4 + threadIdx.x) * 2 + 1];
        sd[0] += a.x; sd[1] += a.y; sd[2] += a.z; sd[3] += a.w;
        qd[0] += b.x; qd[1] += b.y; qd[2] += b.z; qd[3] += b.w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        atomicAdd(st + 4 * threadIdx.x + j, sd[j]);
        atomicAdd(st + y.C + 4 * threadIdx.x + j, qd[j]);
      }
    }
  }
  if (fin.counter && lhn_last_block(fin.counter)) lhn_bn_bwd_finalize_block(fin, sums, save);
}
// dy = A*du + B*y + C with  A = s, B = -s*invstd*dgamma/n, C = -s*dbeta/n + s*invstd*mean*dgamma/n, s = gamma*invstd
__global__ void __launch_bounds__(LHN_FIN_THREADS) k_bn_bwd_finalize(const double* __restrict__ sums, const float* __restrict__ gamma,
                                  const float* __restrict__ save, float* __restrict__ coef, int cs, int coff, int C,
                                  double count, float* __restrict__ dgamma, float* __restrict__ dbeta, float pgrad_scale, int SC) {
  // replica fold as in k_bn_finalize
  __shared__ double part[2 * LHN_FIN_THREADS];
  const int nt = blockDim.x;
  int G = nt / C;
  G = G < 1 ? 1 : (G > LHN_STAT_REPLICAS ? LHN_STAT_REPLICAS : G);
  for (int c0 = 0; c0 < C; c0 += nt) {
    const int g = threadIdx.x / C, c = c0 + (G > 1 ? threadIdx.x - g * C : threadIdx.x);
    if (c0) __syncthreads();
    if (g < G && c < C) {
      double s1 = 0, s2 = 0;
#pragma unroll 4
      for (int r = g; r < LHN_STAT_REPLICAS; r += G) {
        s1 += sums[(size_t)r * 2 * SC + c];
        s2 += sums[(size_t)r * 2 * SC + SC + c];
      }
      part[2 * threadIdx.x] = s1;
      part[2 * threadIdx.x + 1] = s2;
    }
    __syncthreads();
    if (g != 0 || c >= C) continue;
    double db = 0, dg = 0;
    for (int k = 0; k < G; ++k) {
      db += part[2 * (k * C + threadIdx.x)];
      dg += part[2 * (k * C + threadIdx.x) + 1];
    }
    const double mean = save[c], inv = save[SC + c], s = (double)(gamma ? gamma[c] : 1.f) * inv;
    coef[coff + c] = (float)s;
    coef[cs + coff + c] = (float)(-s * inv * dg / count);
    coef[2 * cs + coff + c] = (float)(-s * db / count + s * inv * mean * dg / count);
    if (dgamma) dgamma[c] += (float)dg * pgrad_scale;
    if (dbeta) dbeta[c] += (float)db * pgrad_scale;
  }
}

// out[i] = sum_r part[r][i]
__global__ void __launch_bounds__(256) k_reduce_replicas(float* __restrict__ out, const float* __restrict__ part, int64_t n4,
                                                         int nrep, int64_t rep_stride) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    f4 s = *reinterpret_cast<const f4*>(part + 4 * i);
    for (int r = 1; r < nrep; ++r) s += *reinterpret_cast<const f4*>(part + (size_t)r * rep_stride + 4 * i);
    *reinterpret_cast<f4*>(out + 4 * i) = s;
  }
}

// SyncBatchNorm: fold the replicated sums of one BatchNorm into replica 0 (the others become zero) BEFORE they go on the wire --
// the exchange then moves [2][C] doubles instead of [LHN_STAT_REPLICAS][2][C] (32x fewer bytes over xGMI per BatchNorm), and the
// finalize kernels, which add all replicas, read the global sums unchanged.
__global__ void __launch_bounds__(256) k_fold_stat_replicas(double* __restrict__ st, int64_t n, int nrep) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    double s = st[i];
    for (int r = 1; r < nrep; ++r) {
      s += st[(int64_t)r * n + i];
      st[(int64_t)r * n + i] = 0.0;
    }
    st[i] = s;
  }
}

// ------------------------------------------------------------------ C ABI
extern "C" {

int lhn_fold_stat_replicas(double* stats, int64_t n, int nrep, void* stream) {
  LHN_CHECK_ARG(stats && n > 0 && nrep >= 1, "lhn_fold_stat_replicas: bad argument");
  hipLaunchKernelGGL(k_fold_stat_replicas, dim3((unsigned)((n + 255) / 256 > 64 ? 64 : (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, stats, n, nrep);
  LHN_CHECK_LAUNCH("lhn_fold_stat_replicas");
  return 0;
}

int lhn_reduce_replicas(float* out, const float* part, int64_t n, int nrep, int64_t rep_stride, void* stream) {
  LHN_CHECK_ARG(out && part && n > 0 && n % 4 == 0 && nrep >= 1 && rep_stride % 4 == 0, "lhn_reduce_replicas: bad args (n and stride multiples of 4)");
  hipLaunchKernelGGL(k_reduce_replicas, dim3(grid_cap((n / 4 + 255) / 256, 8)), dim3(256), 0, (hipStream_t)stream, out, part, n / 4, nrep, rep_stride);
  LHN_CHECK_LAUNCH("lhn_reduce_replicas");
  return 0;
}

int lhn_bn_finalize(const double* stats, const float* gamma, const float* beta, float* running_mean, float* running_var,
                    int64_t* nbt, float* table, int cstride, int coff, int C, float* save, double count, float eps,
                    float momentum, float slope, int training, const float* conv_bias, void* stream) {
  return lhn_bn_finalize2(stats, gamma, beta, running_mean, running_var, nbt, table, cstride, coff, C, C, save, count, eps, momentum,
                          slope, training, conv_bias, stream);
}
int lhn_bn_finalize2(const double* stats, const float* gamma, const float* beta, float* running_mean, float* running_var,
                     int64_t* nbt, float* table, int cstride, int coff, int C, int stat_channels, float* save, double count,
                     float eps, float momentum, float slope, int training, const float* conv_bias, void* stream) {
  LHN_CHECK_ARG(table && C > 0 && coff >= 0 && coff + C <= cstride && stat_channels >= C, "lhn_bn_finalize: bad table slice");
  LHN_CHECK_ARG(training ? (stats != nullptr) : (running_mean && running_var), "lhn_bn_finalize: missing statistics");
  LHN_CHECK_ARG(!training || count >= 1, "lhn_bn_finalize: count");
  // training: (channel, replica group) threads, a whole number of channel rows; eval: one thread per channel
  int nt = training ? (C >= LHN_FIN_THREADS ? LHN_FIN_THREADS : (LHN_FIN_THREADS / C > LHN_STAT_REPLICAS ? LHN_STAT_REPLICAS : LHN_FIN_THREADS / C) * C)
                    : (C <= 64 ? 64 : (C <= 128 ? 128 : 256));
  nt = (nt + 63) / 64 * 64;
  hipLaunchKernelGGL(k_bn_finalize, dim3(1), dim3(nt), 0, (hipStream_t)stream, stats,
                     gamma, beta, running_mean, running_var, nbt, table, cstride, coff, C, save, count, eps, momentum,
                     slope, training, conv_bias, stat_channels);
  LHN_CHECK_LAUNCH("lhn_bn_finalize");
  return 0;
}

int lhn_table_fill(float* table, int cstride, int coff, int C, float scale, float shift, float slope, void* stream) {
  LHN_CHECK_ARG(table && C > 0 && coff >= 0 && coff + C <= cstride, "lhn_table_fill: bad table slice");
  hipLaunchKernelGGL(k_table_fill, dim3(1), dim3(128), 0, (hipStream_t)stream, table, cstride, coff, C, scale, shift, slope);
  LHN_CHECK_LAUNCH("lhn_table_fill");
  return 0;
}

int lhn_table_bias(float* table, int cstride, int coff, int C, const float* bias, float slope, void* stream) {
  LHN_CHECK_ARG(table && C > 0 && coff >= 0 && coff + C <= cstride, "lhn_table_bias: bad slice");
  hipLaunchKernelGGL(k_table_bias, dim3(1), dim3(128), 0, (hipStream_t)stream, table, cstride, coff, C, bias, slope);
  LHN_CHECK_LAUNCH("lhn_table_bias");
  return 0;
}

int lhn_ew_fwd(const lhn_view* srcs, int nsrc, const lhn_view* dst, float out_slope, void* stream) {
  return lhn_ew_fwd2(srcs, nsrc, nullptr, dst, out_slope, stream);
}
int lhn_ew_fwd2(const lhn_view* srcs, int nsrc, const float* coef, const lhn_view* dst, float out_slope, void* stream) {
  return lhn_ew_fwd3(srcs, nsrc, coef, dst, out_slope, 0, stream);
}
int lhn_ew_fwd3(const lhn_view* srcs, int nsrc, const float* coef, const lhn_view* dst, float out_slope, int mode, void* stream) {
  LHN_CHECK_ARG(srcs && nsrc >= 1 && nsrc <= 3 && lhn_view_ok(dst) && mode >= 0 && mode <= 3, "lhn_ew_fwd: 1..3 sources, valid dst");
  EwSrcs S;
  memset(&S, 0, sizeof(S));
  S.mode = mode;
  for (int i = 0; i < nsrc; ++i) {
    LHN_CHECK_ARG(lhn_view_ok(&srcs[i]) && srcs[i].C == dst->C && srcs[i].N == dst->N && lhn_no_pend(&srcs[i]), "lhn_ew_fwd: source %d mismatch", i);
    S.v[i] = srcs[i];
    S.coef[i] = coef ? coef[i] : 1.f;
  }
  LHN_CHECK_ARG(dst->C % 4 == 0 && dst->C <= 1024, "lhn_ew_fwd: C=%d", dst->C);
  if (S.mode & 2)
    hipLaunchKernelGGL(k_ew_fwd<true>, dim3(grid_cap((int64_t)dst->N * dst->H, 8)), dim3(256), 0, (hipStream_t)stream, S, nsrc, *dst, out_slope);
  else
    hipLaunchKernelGGL(k_ew_fwd<false>, dim3(grid_cap((int64_t)dst->N * dst->H, 8)), dim3(256), 0, (hipStream_t)stream, S, nsrc, *dst, out_slope);
  LHN_CHECK_LAUNCH("lhn_ew_fwd");
  return 0;
}

// dsrcs[i] == NULL skips source i.  dst_dpool (optional, via the last element of `accumulate`? no) -- see lhn_ew_bwd2.
int lhn_ew_bwd(const lhn_view* srcs, int nsrc, const lhn_view* dst, const float* ddst, float out_slope,
               float* const* dsrcs, const int* accumulate, void* stream) {
  LHN_CHECK_ARG(srcs && nsrc >= 1 && nsrc <= 3 && lhn_view_ok(dst) && ddst && dsrcs && accumulate, "lhn_ew_bwd: bad args");
  for (int i = 0; i < nsrc; ++i) {
    if (!dsrcs[i]) continue;
    const lhn_view* s = &srcs[i];
    LHN_CHECK_ARG(lhn_view_ok(s) && s->C == dst->C, "lhn_ew_bwd: source %d mismatch", i);
    LHN_CHECK_ARG(dst->H % s->H == 0 && dst->W % s->W == 0, "lhn_ew_bwd: non-integer upsample %dx%d -> %dx%d", s->H, s->W, dst->H, dst->W);
    LHN_CHECK_ARG(s->C % 4 == 0 && s->C <= 1024, "lhn_ew_bwd: C=%d", s->C);
    lhn_bnsum nob;
    memset(&nob, 0, sizeof(nob));
    hipLaunchKernelGGL(k_ew_bwd_src, dim3(grid_cap((int64_t)s->N * s->H, 8)), dim3(256), 0, (hipStream_t)stream, *s, *dst, ddst,
                       (const float*)nullptr, out_slope, dsrcs[i], accumulate[i], nob);
  }
  LHN_CHECK_LAUNCH("lhn_ew_bwd");
  return 0;
}
// variant used by the plan: dst may carry a channel-attention gate and its pooled-gradient (dpool)
// a reader-side BatchNorm-sum request is valid for an ungated input view whose channels lie inside the producer's BatchNorm
static int bns_of(const lhn_bnsum* in, const lhn_view* v, lhn_bnsum* out, const char* who) {
  memset(out, 0, sizeof(*out));
  if (!in || !in->sums) return 0;
  LHN_CHECK_ARG(in->save && in->C > 0 && in->coff >= 0 && in->coff % 4 == 0 && in->coff + v->C <= in->C && !v->gate && v->table,
                "%s: BatchNorm sums need an ungated input view with a table, inside the producer's %d channels (offset %d, view %d)", who,
                in->C, in->coff, v->C);
  *out = *in;
  return 0;
}
int lhn_ew_bwd2(const lhn_view* src, const lhn_view* dst, const float* ddst, const float* dst_dpool, float out_slope,
                float* dsrc, int accumulate, void* stream) {
  return lhn_ew_bwd3(src, dst, ddst, dst_dpool, out_slope, dsrc, accumulate, nullptr, stream);
}
// variant used by the plan: dst may carry a channel-attention gate and its pooled-gradient (dpool)
int lhn_ew_bwd3(const lhn_view* src, const lhn_view* dst, const float* ddst, const float* dst_dpool, float out_slope,
                float* dsrc, int accumulate, const lhn_bnsum* bns, void* stream) {
  LHN_CHECK_ARG(lhn_view_ok(src) && lhn_view_ok(dst) && ddst && dsrc && src->C == dst->C && lhn_no_pend(src) && lhn_no_pend(dst), "lhn_ew_bwd2: bad args");
  LHN_CHECK_ARG(dst->H % src->H == 0 && dst->W % src->W == 0, "lhn_ew_bwd2: non-integer upsample");
  LHN_CHECK_ARG((out_slope != LHN_SLOPE_SILU && out_slope != LHN_SLOPE_RELU_SIGMOID) || (dst->H == src->H && dst->W == src->W), "lhn_ew_bwd2: SiLU / ReLU-sigmoid need a single same-size source");
  LHN_CHECK_ARG(src->C % 4 == 0 && src->C <= 1024, "lhn_ew_bwd2: C=%d", src->C);
  lhn_bnsum bs;
  if (bns_of(bns, src, &bs, "lhn_ew_bwd3")) return 1;
  hipLaunchKernelGGL(k_ew_bwd_src, dim3(grid_cap((int64_t)src->N * src->H, 8)), dim3(256), 0, (hipStream_t)stream, *src, *dst, ddst,
                     dst_dpool, out_slope, dsrc, accumulate, bs);
  LHN_CHECK_LAUNCH("lhn_ew_bwd2");
  return 0;
}

// up to three sources of dst's own resolution in one pass (see k_ew_bwd_multi)
int lhn_ew_bwd_multi(const lhn_view* srcs, int nsrc, const lhn_view* dst, const float* ddst, const float* dst_dpool, float out_slope,
                     float* const* dsrcs, const int* accumulate, const lhn_bnsum* const* bns, void* stream) {
  LHN_CHECK_ARG(srcs && nsrc >= 1 && nsrc <= 3 && lhn_view_ok(dst) && ddst && dsrcs && accumulate && lhn_no_pend(dst), "lhn_ew_bwd_multi: bad args");
  LHN_CHECK_ARG(out_slope != LHN_SLOPE_SILU && out_slope != LHN_SLOPE_RELU_SIGMOID, "lhn_ew_bwd_multi: SiLU / ReLU-sigmoid take lhn_ew_bwd3");
  LHN_CHECK_ARG(dst->C % 4 == 0 && dst->C <= 1024, "lhn_ew_bwd_multi: C=%d", dst->C);
  EwBwdMulti m;
  memset(&m, 0, sizeof(m));
  m.n = nsrc;
  for (int k = 0; k < nsrc; ++k) {
    const lhn_view* v = &srcs[k];
    LHN_CHECK_ARG(lhn_view_ok(v) && lhn_no_pend(v) && dsrcs[k] && v->C == dst->C && v->N == dst->N && v->H == dst->H && v->W == dst->W,
                  "lhn_ew_bwd_multi: source %d must have the destination's geometry", k);
    m.src[k] = *v;
    m.dsrc[k] = dsrcs[k];
    m.acc[k] = accumulate[k];
    if (bns_of(bns ? bns[k] : nullptr, v, &m.bs[k], "lhn_ew_bwd_multi")) return 1;
  }
  hipLaunchKernelGGL(k_ew_bwd_multi, dim3(grid_cap((int64_t)dst->N * dst->H, 8)), dim3(256), 0, (hipStream_t)stream, m, *dst, ddst, dst_dpool, out_slope);
  LHN_CHECK_LAUNCH("lhn_ew_bwd_multi");
  return 0;
}

int lhn_maxpool2_fwd(const lhn_view* x, const lhn_view* y, void* stream) {
  LHN_CHECK_ARG(lhn_view_ok(x) && lhn_view_ok(y) && x->C == y->C, "lhn_maxpool2_fwd: bad views");
  LHN_CHECK_ARG(y->H == (x->H + 1) / 2 && y->W == (x->W + 1) / 2 && y->N == x->N, "lhn_maxpool2_fwd: geometry");
  LHN_CHECK_ARG(y->C % 4 == 0 && y->C <= 1024, "lhn_maxpool2_fwd: C=%d", y->C);
  LHN_CHECK_ARG(lhn_no_pend(x), "lhn_maxpool2_fwd: lhn_view.pend is reserved (NULL)");
  hipLaunchKernelGGL(k_maxpool2_fwd, dim3(grid_cap((int64_t)y->N * y->H, 8)), dim3(256), 0, (hipStream_t)stream, *x, *y);
  LHN_CHECK_LAUNCH("lhn_maxpool2_fwd");
  return 0;
}
int lhn_maxpool2_bwd(const lhn_view* x, const lhn_view* y, const float* dy, float* dx, int dx_accumulate, void* stream) {
  return lhn_maxpool2_bwd2(x, y, dy, dx, dx_accumulate, nullptr, stream);
}
int lhn_maxpool2_bwd2(const lhn_view* x, const lhn_view* y, const float* dy, float* dx, int dx_accumulate, const lhn_bnsum* bns,
                      void* stream) {
  return lhn_maxpool2_bwd3(x, y, dy, dx, dx_accumulate, bns, nullptr, stream);
}
int lhn_maxpool2_bwd3(const lhn_view* x, const lhn_view* y, const float* dy, float* dx, int dx_accumulate, const lhn_bnsum* bns,
                      const lhn_grad_adds* adds, void* stream) {
  LHN_CHECK_ARG(lhn_view_ok(x) && lhn_view_ok(y) && dy && dx && x->C == y->C && lhn_no_pend(x), "lhn_maxpool2_bwd: bad args");
  LHN_CHECK_ARG(y->C % 4 == 0 && y->C <= 1024, "lhn_maxpool2_bwd: C=%d", y->C);
  lhn_bnsum bs;
  if (bns_of(bns, x, &bs, "lhn_maxpool2_bwd2")) return 1;
  lhn_grad_adds ad;
  memset(&ad, 0, sizeof(ad));
  if (adds && (adds->same || adds->pooled)) {
    ad = *adds;
    LHN_CHECK_ARG(x->H % 2 == 0 && x->W % 2 == 0 && y->H == x->H / 2 && y->W == x->W / 2, "lhn_maxpool2_bwd3: gradient addends need even H, W (%dx%d)", x->H, x->W);
    LHN_CHECK_ARG(!ad.same || (ad.same_cstride % 4 == 0 && ad.same_coff % 4 == 0 && ad.same_coff >= 0 && ad.same_coff + x->C <= ad.same_cstride),
                  "lhn_maxpool2_bwd3: same-resolution addend layout");
    LHN_CHECK_ARG(!ad.pooled || (ad.OH > 0 && ad.OW > 0 && ad.OH <= x->H && ad.OW <= x->W && ad.pooled_cstride % 4 == 0 && ad.pooled_coff % 4 == 0 &&
                                 ad.pooled_coff >= 0 && ad.pooled_coff + x->C <= ad.pooled_cstride),
                  "lhn_maxpool2_bwd3: pooled addend layout");
    hipLaunchKernelGGL(k_maxpool2_bwd<true>, dim3(grid_cap((int64_t)y->N * y->H, 8)), dim3(256), 0, (hipStream_t)stream, *x, *y, dy, dx, dx_accumulate, bs, ad);
  } else {
    hipLaunchKernelGGL(k_maxpool2_bwd<false>, dim3(grid_cap((int64_t)y->N * y->H, 8)), dim3(256), 0, (hipStream_t)stream, *x, *y, dy, dx, dx_accumulate, bs, ad);
  }
  LHN_CHECK_LAUNCH("lhn_maxpool2_bwd");
  return 0;
}

int lhn_avgpool_fwd(const lhn_view* x, float* out, int OH, int OW, void* stream) {
  return lhn_avgpool_fwd2(x, out, OH, OW, x ? x->C : 0, 0, stream);
}
int lhn_avgpool_fwd2(const lhn_view* x, float* out, int OH, int OW, int out_cstride, int out_coff, void* stream) {
  LHN_CHECK_ARG(lhn_view_ok(x) && out && OH > 0 && OW > 0 && out_coff >= 0 && out_coff % 4 == 0 && out_cstride % 4 == 0 &&
                    out_coff + x->C <= out_cstride,
                "lhn_avgpool_fwd: bad args");
  LHN_CHECK_ARG(x->C % 4 == 0 && x->C <= 1024, "lhn_avgpool_fwd: C=%d", x->C);
  LHN_CHECK_ARG(lhn_no_pend(x), "lhn_avgpool_fwd: lhn_view.pend is reserved (NULL)");
  const int binmax = ((x->H + OH - 1) / OH + 1) * ((x->W + OW - 1) / OW + 1);
  if (binmax <= 25) {          // bins of at most 4x4 (+1: adaptive bins may overlap by a pixel)
    const int64_t total = (int64_t)x->N * OH * OW * (x->C / 4);
    hipLaunchKernelGGL(k_avgpool_small, dim3(grid_cap((total + 255) / 256, 16)), dim3(256), 0, (hipStream_t)stream, *x, out, OH, OW, out_cstride, out_coff);
  } else
  {
    BnSlices sl;
    memset(&sl, 0, sizeof(sl));
    lhn_view nosrc;
    memset(&nosrc, 0, sizeof(nosrc));
    hipLaunchKernelGGL((k_avgpool_fwd<false, false>), dim3(x->N * OH * OW), dim3(256), 0, (hipStream_t)stream, *x, out, OH, OW, out_cstride, out_coff,
                       (float*)nullptr, sl, nosrc);
  }
  LHN_CHECK_LAUNCH("lhn_avgpool_fwd");
  return 0;
}
static int mk_slices(const lhn_bn_slices* in, int C, BnSlices* sl, const char* who) {
  memset(sl, 0, sizeof(*sl));
  if (!in) return 0;
  LHN_CHECK_ARG(in->n >= 0 && in->n <= 2, "%s: 0..2 BatchNorm slices", who);
  sl->n = in->n;
  for (int k = 0; k < in->n; ++k) {
    LHN_CHECK_ARG(in->save[k] && in->lo[k] >= 0 && in->C[k] > 0 && in->lo[k] % 4 == 0 && in->C[k] % 4 == 0 && in->lo[k] + in->C[k] <= C,
                  "%s: BatchNorm slice %d = channels [%d, %d) of %d", who, k, in->lo[k], in->lo[k] + in->C[k], C);
    sl->save[k] = in->save[k];
    sl->sums[k] = in->sums[k];
    sl->lo[k] = in->lo[k];
    sl->C[k] = in->C[k];
  }
  return 0;
}
int lhn_avgpool_fwd3(const lhn_view* x, float* out, int OH, int OW, float* pstat, const lhn_bn_slices* slices, void* stream) {
  return lhn_avgpool_fwd4(x, out, OH, OW, pstat, slices, nullptr, stream);
}
int lhn_avgpool_fwd4(const lhn_view* x, float* out, int OH, int OW, float* pstat, const lhn_bn_slices* slices, const lhn_view* copy_src,
                     void* stream) {
  if (!pstat && !copy_src) return lhn_avgpool_fwd2(x, out, OH, OW, x ? x->C : 0, 0, stream);
  LHN_CHECK_ARG(lhn_view_ok(x) && out && OH > 0 && OW > 0 && x->C % 4 == 0 && x->C <= 1024 && x->coff == 0 && x->C == x->cstride && lhn_no_pend(x),
                "lhn_avgpool_fwd4: the statistics / copy forms pool a whole buffer");
  LHN_CHECK_ARG(!copy_src || (lhn_view_ok(copy_src) && lhn_no_pend(copy_src) && copy_src->N == x->N && copy_src->H == x->H && copy_src->W == x->W &&
                              copy_src->C < x->C && copy_src->data != x->data),
                "lhn_avgpool_fwd4: copy_src = the first channels of the pooled tensor, same pixels, another buffer");
  BnSlices sl;
  if (mk_slices(pstat ? slices : nullptr, x->C, &sl, "lhn_avgpool_fwd4")) return 1;
  lhn_view src;
  memset(&src, 0, sizeof(src));
  if (copy_src) src = *copy_src;
  const dim3 g(x->N * OH * OW), bk(256);
  hipStream_t s = (hipStream_t)stream;
  if (pstat && copy_src) hipLaunchKernelGGL((k_avgpool_fwd<true, true>), g, bk, 0, s, *x, out, OH, OW, x->C, 0, pstat, sl, src);
  else if (pstat) hipLaunchKernelGGL((k_avgpool_fwd<true, false>), g, bk, 0, s, *x, out, OH, OW, x->C, 0, pstat, sl, src);
  else hipLaunchKernelGGL((k_avgpool_fwd<false, true>), g, bk, 0, s, *x, out, OH, OW, x->C, 0, pstat, sl, src);
  LHN_CHECK_LAUNCH("lhn_avgpool_fwd4");
  return 0;
}
int lhn_avgpool_bwd(const lhn_view* x, const float* dout, int OH, int OW, float* dx, int dx_accumulate, void* stream) {
  return lhn_avgpool_bwd2(x, dout, OH, OW, x ? x->C : 0, 0, dx, dx_accumulate, stream);
}
int lhn_avgpool_bwd2(const lhn_view* x, const float* dout, int OH, int OW, int out_cstride, int out_coff, float* dx,
                     int dx_accumulate, void* stream) {
  return lhn_avgpool_bwd3(x, dout, OH, OW, out_cstride, out_coff, dx, dx_accumulate, nullptr, stream);
}
int lhn_avgpool_bwd3(const lhn_view* x, const float* dout, int OH, int OW, int out_cstride, int out_coff, float* dx,
                     int dx_accumulate, const lhn_bnsum* bns, void* stream) {
  LHN_CHECK_ARG(lhn_view_ok(x) && dout && dx && out_coff >= 0 && out_coff + x->C <= out_cstride, "lhn_avgpool_bwd: bad args");
  LHN_CHECK_ARG(x->C % 4 == 0 && x->C <= 1024, "lhn_avgpool_bwd: C=%d", x->C);
  lhn_bnsum bs;
  if (bns_of(bns, x, &bs, "lhn_avgpool_bwd3")) return 1;
  hipLaunchKernelGGL(k_avgpool_bwd, dim3(grid_cap((int64_t)x->N * x->H, 8)), dim3(256), 0, (hipStream_t)stream, *x, dout, OH, OW, dx, dx_accumulate, out_cstride, out_coff, bs);
  LHN_CHECK_LAUNCH("lhn_avgpool_bwd");
  return 0;
}

int lhn_ca_mlp_fwd(const float* pooled, const float* w3, const float* gamma, const float* beta, float* rmean, float* rvar,
                   int64_t* nbt, const float* w1, const float* b1, const float* w2, const float* b2, const float* dropmask,
                   float* gate, int gate_stride, int gate_coff, float* save, int N, int C, float eps, float momentum,
                   int training, int stage, double* gsum, double count_scale, void* stream) {
  LHN_CHECK_ARG(pooled && w3 && w1 && b1 && w2 && b2 && gate && save, "lhn_ca_mlp_fwd: null pointer");
  LHN_CHECK_ARG(gamma ? (beta && rmean && rvar) : !training, "lhn_ca_mlp_fwd: BatchNorm tensors missing (gamma NULL = deployed form, eval only)");
  LHN_CHECK_ARG(C > 0 && C <= 256 && C % 2 == 0 && N > 0, "lhn_ca_mlp_fwd: C=%d (<=256)", C);
  hipStream_t s = (hipStream_t)stream;
  LHN_CHECK_ARG(stage == 0 || (gsum && stage >= 1 && stage <= 2 && count_scale >= 1), "lhn_ca_mlp_fwd: stage %d needs gsum", stage);
  // (one launch for both kernels -- every workgroup recomputing the batch statistics from the pooled tensor -- was measured
  // slower than the two launches: forward of the MSRB hourglass 2.770 vs 2.714 ms, round 3)
  hipLaunchKernelGGL(k_ca1, dim3((C + 31) / 32), dim3(1024), 0, s, pooled, w3, gamma, beta, rmean, rvar, nbt, dropmask, save, N, C, eps, momentum, training, stage, gsum, stage ? count_scale : 1.0);
  if (stage != 1) hipLaunchKernelGGL(k_ca2, dim3(N), dim3(128), 0, s, w1, b1, w2, b2, save, gate, gate_stride, gate_coff, N, C);
  LHN_CHECK_LAUNCH("lhn_ca_mlp_fwd");
  return 0;
}

int lhn_gate_bwd_reduce(const lhn_view* y, const float* dz, float* dgate, void* stream) {
  return lhn_gate_bwd_reduce2(y, dz, dgate, nullptr, nullptr, stream);
}
int lhn_gate_bwd_reduce2(const lhn_view* y, const float* dz, float* dgate, float* tsum, const lhn_bn_slices* slices, void* stream) {
  return lhn_gate_bwd_reduce3(y, dz, dgate, tsum, slices, 0, stream);
}
int lhn_gate_bwd_reduce3(const lhn_view* y, const float* dz, float* dgate, float* tsum, const lhn_bn_slices* slices, int prezeroed,
                         void* stream) {
  LHN_CHECK_ARG(lhn_view_ok(y) && dz && dgate && y->C % 4 == 0 && y->C <= 1024 && lhn_no_pend(y), "lhn_gate_bwd_reduce: bad args");
  LHN_CHECK_ARG(!tsum || (y->coff == 0 && y->C == y->cstride && tsum == dgate + (size_t)y->N * y->C),
                "lhn_gate_bwd_reduce2: tsum = the 2*N*C floats behind dgate, whole-buffer view");
  hipStream_t s = (hipStream_t)stream;
  BnSlices sl;
  if (mk_slices(tsum ? slices : nullptr, y->C, &sl, "lhn_gate_bwd_reduce2")) return 1;
  // (prezeroed: the caller zeroed dgate / tsum already -- the plan keeps them in the arena its backward zeroes with ONE memset;
  // eight attentions of variant B cost eight 4.8 us fill launches per step otherwise)
  if (!prezeroed && hipMemsetAsync(dgate, 0, (size_t)y->N * y->C * 4 * (tsum ? 3 : 1), s) != hipSuccess) {
    lhn_set_error("lhn_gate_bwd_reduce: memset failed");
    return 2;
  }
  int split = (y->H * y->W + 127) / 128;
  if (split < 1) split = 1;
  if (split > 32) split = 32;
  if (lhn_deterministic_mode()) split = 1;          // one workgroup per image: a single (ordered) writer per d(gate) element
  hipLaunchKernelGGL(k_gate_bwd_reduce, dim3(y->N, split), dim3(256), 0, s, *y, dz, dgate, tsum, sl);
  LHN_CHECK_LAUNCH("lhn_gate_bwd_reduce");
  return 0;
}

// scratch: dahat [N][C] floats taken from the tail of `dpool`'s owner?  No: caller passes save; dahat reuses
// the `a`-sized region appended after the forward save area (save must have room for N*C extra floats).
int lhn_ca_mlp_bwd(const float* pooled, const float* w3, const float* gamma, const float* w1, const float* w2,
                   const float* dropmask, const float* save, const float* dgate, float* dpool, int cstride, int coff, int H,
                   int W, float* dw3, float* dgamma, float* dbeta, float* dw1, float* db1, float* dw2, float* db2, int N,
                   int C, int stage, double* gsum, double count_scale, float pgrad_scale, void* stream) {
  return lhn_ca_mlp_bwd2(pooled, w3, gamma, w1, w2, dropmask, save, dgate, dpool, cstride, coff, H, W, dw3, dgamma, dbeta, dw1, db1, dw2,
                         db2, N, C, stage, gsum, count_scale, pgrad_scale, nullptr, nullptr, nullptr, stream);
}
int lhn_ca_mlp_bwd2(const float* pooled, const float* w3, const float* gamma, const float* w1, const float* w2,
                    const float* dropmask, const float* save, const float* dgate, float* dpool, int cstride, int coff, int H,
                    int W, float* dw3, float* dgamma, float* dbeta, float* dw1, float* db1, float* dw2, float* db2, int N,
                    int C, int stage, double* gsum, double count_scale, float pgrad_scale, const float* tsum, const float* pstat,
                    const lhn_bn_slices* slices, void* stream) {
  LHN_CHECK_ARG(!tsum == !pstat && (!tsum || (slices && coff == 0 && C == cstride)), "lhn_ca_mlp_bwd2: tsum, pstat and slices come together (whole buffer)");
  BnSlices sl;
  if (mk_slices(tsum ? slices : nullptr, cstride, &sl, "lhn_ca_mlp_bwd2")) return 1;
  LHN_CHECK_ARG(pooled && w3 && gamma && w1 && w2 && save && dgate && dpool && dw3 && dgamma && dbeta && dw1 && db1 && dw2 && db2,
                "lhn_ca_mlp_bwd: null pointer");
  LHN_CHECK_ARG(C > 0 && C <= 256 && C % 2 == 0, "lhn_ca_mlp_bwd: C=%d", C);
  hipStream_t s = (hipStream_t)stream;
  // scratch behind the forward's save area: dahat [N][C] | dv [N][C] | dh [N][C/2]  (save holds 6 N C + 2 C floats in all)
  float* dahat = const_cast<float*>(save) + (int64_t)N * C * 3 + (int64_t)N * (C / 2) + 2 * C;
  float* dvbuf = dahat + (int64_t)N * C;
  float* dhbuf = dvbuf + (int64_t)N * C;
  LHN_CHECK_ARG(stage == 0 || (gsum && stage >= 1 && stage <= 2 && count_scale >= 1), "lhn_ca_mlp_bwd: stage %d needs gsum", stage);
  if (stage != 2) hipLaunchKernelGGL(k_ca_bwd2, dim3(N), dim3(256), 0, s, w1, w2, save, dgate, dahat, dvbuf, dhbuf, N, C);
  hipLaunchKernelGGL(k_ca_bwd1, dim3((C + 31) / 32 + (C * C + C + C / 2 + 1023) / 1024), dim3(1024), 0, s, pooled, w3, gamma, dropmask, save, dahat, dpool, cstride, coff, H, W, dw3, dgamma, dbeta, N, C, 1, stage, gsum, stage ? count_scale : 1.0, stage ? pgrad_scale : 1.f, tsum, pstat, sl, dvbuf, dhbuf, dw1, db1, dw2, db2);
  LHN_CHECK_LAUNCH("lhn_ca_mlp_bwd");
  return 0;
}

int lhn_bn_bwd_reduce(const lhn_view* y, const lhn_gradview* g, const float* save, double* sums, const lhn_bnbwdfin* finp,
                      void* stream) {
  lhn_bnbwdfin fin;
  if (finp) fin = *finp; else fin.counter = nullptr;
  LHN_CHECK_ARG(lhn_view_ok(y) && g && g->dz && save && sums && y->C % 4 == 0 && y->C <= 1024 && lhn_no_pend(y), "lhn_bn_bwd_reduce: bad args");
  hipLaunchKernelGGL(k_bn_bwd_reduce, dim3(grid_cap((int64_t)y->N * y->H, 8)), dim3(256), 0, (hipStream_t)stream, *y, *g, save, sums, fin);
  LHN_CHECK_LAUNCH("lhn_bn_bwd_reduce");
  return 0;
}
int lhn_bn_bwd_finalize(const double* sums, const float* gamma, const float* save, float* coef, int cstride, int coff, int C,
                        double count, float* dgamma, float* dbeta, float pgrad_scale, void* stream) {
  return lhn_bn_bwd_finalize2(sums, gamma, save, coef, cstride, coff, C, C, count, dgamma, dbeta, pgrad_scale, stream);
}
int lhn_bn_bwd_finalize2(const double* sums, const float* gamma, const float* save, float* coef, int cstride, int coff, int C,
                         int stat_channels, double count, float* dgamma, float* dbeta, float pgrad_scale, void* stream) {
  LHN_CHECK_ARG(sums && save && coef && C > 0 && coff + C <= cstride && stat_channels >= C, "lhn_bn_bwd_finalize: bad args");
  int nt = C >= LHN_FIN_THREADS ? LHN_FIN_THREADS : (LHN_FIN_THREADS / C > LHN_STAT_REPLICAS ? LHN_STAT_REPLICAS : LHN_FIN_THREADS / C) * C;
  nt = (nt + 63) / 64 * 64;
  hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(1), dim3(nt), 0, (hipStream_t)stream, sums, gamma, save, coef, cstride, coff, C, count, dgamma, dbeta, pgrad_scale, stat_channels);
  LHN_CHECK_LAUNCH("lhn_bn_bwd_finalize");
  return 0;
}
}  // extern "C"

// ------------------------------------------------------------------ Adam over ONE flat parameter buffer
// torch.optim.Adam (dist_train.py:64-69: Adam, default betas / eps, no weight decay, no amsgrad) in the arithmetic order of
// torch's single-tensor form:  m = m + (1 - b1) (g - m);  v = b2 v + (1 - b2) g g;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps).
// torch's fused kernel walks a single tensor in 65,536-element chunks, one workgroup each: 5 workgroups and 41 us for the 289 k
// parameters of variant B; this is a plain streaming kernel.
__global__ void __launch_bounds__(256) k_adam_flat(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n4, int64_t n, float omb1, float b2, float omb2,
                                                   float eps, float wd, float step_size, float bc2_sqrt) {
  // omb1 = 1 - beta1, omb2 = 1 - beta2 rounded from the DOUBLE difference, as torch forms them (1 - 0.999f is 4.7e-5 off 0.001)
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f4 pp = *reinterpret_cast<const f4*>(p + 4 * i), gg = *reinterpret_cast<const f4*>(g + 4 * i);
    f4 mm = *reinterpret_cast<const f4*>(m + 4 * i), vv = *reinterpret_cast<const f4*>(v + 4 * i);
    if (wd != 0.f) gg += pp * wd;
    mm += (gg - mm) * omb1;
    vv = vv * b2 + gg * gg * omb2;
    const f4 den = (f4){sqrtf(vv.x) / bc2_sqrt + eps, sqrtf(vv.y) / bc2_sqrt + eps, sqrtf(vv.z) / bc2_sqrt + eps, sqrtf(vv.w) / bc2_sqrt + eps};
    pp -= (f4){mm.x / den.x, mm.y / den.y, mm.z / den.z, mm.w / den.w} * step_size;
    *reinterpret_cast<f4*>(p + 4 * i) = pp;
    *reinterpret_cast<f4*>(m + 4 * i) = mm;
    *reinterpret_cast<f4*>(v + 4 * i) = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {      // tail (n % 4 elements)
    const int64_t i = 4 * n4 + threadIdx.x;
    float gg = g[i];
    if (wd != 0.f) gg += p[i] * wd;
    const float mm = m[i] + (gg - m[i]) * omb1, vv = v[i] * b2 + gg * gg * omb2;
    p[i] -= mm / (sqrtf(vv) / bc2_sqrt + eps) * step_size;
    m[i] = mm;
    v[i] = vv;
  }
}

extern "C" int lhn_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1,
                             double beta2, double eps, double weight_decay, int64_t step, void* stream) {
  LHN_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "lhn_adam_step: null pointer / n / step");
  LHN_CHECK_ARG(((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) |
                  reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) == 0, "lhn_adam_step: buffers must be 16-byte aligned");
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);      // python floats in torch
  const int64_t n4 = n / 4;
  int64_t grid = (n4 + 255) / 256;
  const int64_t cap = (int64_t)lhn_num_cus() * 8;
  if (grid > cap) grid = cap;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(k_adam_flat, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n4, n, (float)(1.0 - beta1),
                     (float)beta2, (float)(1.0 - beta2), (float)eps, (float)weight_decay, (float)(lr / bc1), (float)sqrt(bc2));
  LHN_CHECK_LAUNCH("lhn_adam_step");
  return 0;
}

