// Gaussian heatmap encode / decode / balanced-MSE loss kernels (HBM-bound, NCHW [N,K,H,W] fp32).
// Reference arithmetic replaced: see include/lhn.h.
#include "lhn_common.h"

// The decode / encode arithmetic must round like the reference's numpy float32 expressions: no fused
// multiply-add contraction anywhere in this file (HIP's own *_rn helpers are header inlines that still
// contract, so the helpers below are defined under contract(off); the file is also built with -ffp-contract=off).
#pragma clang fp contract(off)
__device__ __forceinline__ float mul_rn(float a, float b) { return a * b; }
__device__ __forceinline__ float add_rn(float a, float b) { return a + b; }
__device__ __forceinline__ float sub_rn(float a, float b) { return a - b; }

// ------------------------------------------------------------------ encode
// generateTarget.py:100-123 (unbiased) computes exp() in float64 (numpy>=2 promotion of a float32
// array with a float64 numpy scalar) and rounds to float32 on assignment; :125-154 (biased) works
// in float32 on an integer-centred (6*sigma+1)^2 patch.
__global__ void __launch_bounds__(256) k_encode(const float* __restrict__ joints, const float* __restrict__ visible,
                                                float* __restrict__ target, float* __restrict__ weight, int NK, int H, int W,
                                                double stride_x, double stride_y, float sigma, int unbiased) {
  const int nk = blockIdx.x;
  if (nk >= NK) return;
  const float jx = joints[nk * 3 + 0], jy = joints[nk * 3 + 1];
  float wgt = visible[nk * 3 + 0];
  const double r = (double)sigma * 3.0;
  float* out = target + (int64_t)nk * H * W;
  const int HW = H * W;
  if (unbiased == 1) {
    const double mx = (double)jx / stride_x, my = (double)jy / stride_y;
    if (mx - r >= W || my - r >= H || mx + r + 1 < 0 || my + r + 1 < 0) wgt = 0.f;
    const bool on = wgt > 0.5f;
    const double den = 2.0 * (double)sigma * (double)sigma;
    for (int i = threadIdx.x * 4; i < HW; i += blockDim.x * 4) {
      f4 v = (f4){0.f, 0.f, 0.f, 0.f};
      if (on) {
        float t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int p = i + j;
          const double dx = (double)(float)(p % W) - mx, dy = (double)(float)(p / W) - my;
          t[j] = (float)exp(-(dx * dx + dy * dy) / den);
        }
        v = (f4){t[0], t[1], t[2], t[3]};
      }
      *reinterpret_cast<f4*>(out + i) = v;
    }
  } else {
    const int mx = (int)((double)jx / stride_x + 0.5), my = (int)((double)jy / stride_y + 0.5);
    const int ri = (int)r;
    const int ulx = (int)((double)mx - r), uly = (int)((double)my - r);
    const int brx = (int)((double)mx + r + 1), bry = (int)((double)my + r + 1);
    if (ulx >= W || uly >= H || brx < 0 || bry < 0) wgt = 0.f;
    const bool on = wgt > 0.5f;
    const float den = 2.f * sigma * sigma;
    const int size = 2 * ri + 1, c0 = size / 2;
    // UDP (generateTarget.py:160-236, stride_* = (image-1)/(heatmap-1)): same patch, but the Gaussian is centred on the
    // sub-pixel position  size//2 + (joint/stride - mu)  and evaluated in float64 like the reference's mixed-type expression
    const double x0 = (double)c0 + ((double)jx / stride_x - (double)mx), y0 = (double)c0 + ((double)jy / stride_y - (double)my);
    for (int i = threadIdx.x; i < HW; i += blockDim.x) {
      const int x = i % W, y = i / W;
      float v = 0.f;
      if (on && x >= ulx && x < brx && y >= uly && y < bry) {
        if (unbiased == 2) {
          const double gx = (double)(float)(x - ulx) - x0, gy = (double)(float)(y - uly) - y0;
          v = (float)exp(-(gx * gx + gy * gy) / (2.0 * (double)sigma * (double)sigma));
        } else {
          const float gx = (float)(x - ulx) - (float)c0, gy = (float)(y - uly) - (float)c0;
          const float a = add_rn(mul_rn(gx, gx), mul_rn(gy, gy));
          v = expf(-a / den);
        }
      }
      out[i] = v;
    }
  }
  if (threadIdx.x == 0) weight[nk] = wgt;
}

// ------------------------------------------------------------------ argmax (first max wins, NaN is max)
struct MaxI {
  float v;
  int i;
};
__device__ __forceinline__ bool better(float av, int ai, float bv, int bi) {
  const bool an = av != av, bn = bv != bv;
  if (an || bn) return an && (!bn || ai < bi);
  return av > bv || (av == bv && ai < bi);
}
__device__ __forceinline__ MaxI block_argmax(const float* __restrict__ m, int HW) {
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = threadIdx.x * 4; i < HW; i += blockDim.x * 4) {
    const f4 q = *reinterpret_cast<const f4*>(m + i);
    const float e[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (better(e[j], i + j, bv, bi)) {
        bv = e[j];
        bi = i + j;
      }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (better(ov, oi, bv, bi)) {
      bv = ov;
      bi = oi;
    }
  }
  __shared__ float sv[4];
  __shared__ int si[4];
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sv[wv] = bv;
    si[wv] = bi;
  }
  __syncthreads();
  MaxI r{sv[0], si[0]};
  for (int k = 1; k < (int)(blockDim.x >> 6); ++k)
    if (better(sv[k], si[k], r.v, r.i)) {
      r.v = sv[k];
      r.i = si[k];
    }
  return r;
}

__device__ __forceinline__ float sgn(float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : (d == d ? 0.f : d)); }

// mode 0: top_down_eval.py:440-452 ; mode 1: heatmap_post_processing.py:6-33 (clamped, then +0.5)
__device__ __forceinline__ void refine_xy(const float* __restrict__ m, int H, int W, int mode, float& x, float& y) {
  const int px = (int)x, py = (int)y;
  if (mode == 0) {
    if (1 < px && px < W - 1 && 1 < py && py < H - 1) {
      const float dx = m[py * W + px + 1] - m[py * W + px - 1];
      const float dy = m[(py + 1) * W + px] - m[(py - 1) * W + px];
      x = add_rn(x, sgn(dx) * 0.25f);
      y = add_rn(y, sgn(dy) * 0.25f);
    }
  } else {
    const int xr = min(px + 1, W - 1), xl = max(px - 1, 0), yd = min(py + 1, H - 1), yu = max(py - 1, 0);
    x = add_rn(x, m[py * W + xr] > m[py * W + xl] ? 0.25f : -0.25f);
    y = add_rn(y, m[yd * W + px] > m[yu * W + px] ? 0.25f : -0.25f);
    x = add_rn(x, 0.5f);
    y = add_rn(y, 0.5f);
  }
}
// post_transforms.py:35-46, float32 left-to-right, no contraction
__device__ __forceinline__ void xform_xy(float x, float y, const float* c, const float* s, int W, int H, int udp,
                                         float& ox, float& oy) {
  const float sw = mul_rn(s[0], 200.f), sh = mul_rn(s[1], 200.f);
  const float sx = udp ? sw / ((float)W - 1.f) : sw / (float)W;
  const float sy = udp ? sh / ((float)H - 1.f) : sh / (float)H;
  ox = sub_rn(add_rn(mul_rn(x, sx), c[0]), mul_rn(sw, 0.5f));
  oy = sub_rn(add_rn(mul_rn(y, sy), c[1]), mul_rn(sh, 0.5f));
}

__global__ void __launch_bounds__(256) k_decode(const float* __restrict__ hm, const float* __restrict__ center,
                                                const float* __restrict__ scale, float* __restrict__ hm_preds,
                                                float* __restrict__ preds, float* __restrict__ maxvals,
                                                int32_t* __restrict__ index, int K, int H, int W, int post) {
  const int nk = blockIdx.x;
  const float* m = hm + (int64_t)nk * H * W;
  const MaxI r = block_argmax(m, H * W);
  if (threadIdx.x == 0) {
    float x = (float)(r.i % W), y = (float)(r.i / W);
    if (!(r.v > 0.f)) x = y = -1.f;
    if (post == 1) refine_xy(m, H, W, 0, x, y);
    if (maxvals) maxvals[nk] = r.v;
    if (index) index[nk] = r.i;
    if (hm_preds) {
      hm_preds[nk * 2 + 0] = x;
      hm_preds[nk * 2 + 1] = y;
    }
    if (preds) {
      const int n = nk / K;
      float ox, oy;
      xform_xy(x, y, center + n * 2, scale + n * 2, W, H, 0, ox, oy);
      preds[nk * 2 + 0] = ox;
      preds[nk * 2 + 1] = oy;
    }
  }
}

__global__ void k_refine(const float* __restrict__ hm, float* __restrict__ preds, int NK, int H, int W, int mode) {
  const int nk = blockIdx.x * blockDim.x + threadIdx.x;
  if (nk >= NK) return;
  float x = preds[nk * 2], y = preds[nk * 2 + 1];
  refine_xy(hm + (int64_t)nk * H * W, H, W, mode, x, y);
  preds[nk * 2] = x;
  preds[nk * 2 + 1] = y;
}

__global__ void k_transform(const float* __restrict__ coords, const float* __restrict__ center,
                            const float* __restrict__ scale, float* __restrict__ out, int NK, int K, int W, int H, int udp) {
  const int nk = blockIdx.x * blockDim.x + threadIdx.x;
  if (nk >= NK) return;
  const int n = nk / K;
  float ox, oy;
  xform_xy(coords[nk * 2], coords[nk * 2 + 1], center + n * 2, scale + n * 2, W, H, udp, ox, oy);
  out[nk * 2] = ox;
  out[nk * 2 + 1] = oy;
}

// ------------------------------------------------------------------ k x k max-pool peak keep (separable)
// HeatmapParser.py:41-50: h *= (maxpool_kxk(h) == h); -inf padding as in F.max_pool2d.
__global__ void __launch_bounds__(256) k_nms(float* __restrict__ hm, float* __restrict__ scratch, int H, int W, int k) {
  const int nk = blockIdx.x, p = k / 2, HW = H * W;
  float* m = hm + (int64_t)nk * HW;
  float* s = scratch + (int64_t)nk * HW;
  for (int i = threadIdx.x; i < HW; i += blockDim.x) {
    const int x = i % W, y = i / W;
    float mx = -INFINITY;
    for (int d = -p; d <= p; ++d) {
      const int xx = x + d;
      if (xx >= 0 && xx < W) mx = fmaxf(mx, m[y * W + xx]);
    }
    s[i] = mx;
  }
  __syncthreads();  // one block owns the whole map; pass 2 reads only `s` and its own m[i]
  for (int i = threadIdx.x; i < HW; i += blockDim.x) {
    const int x = i % W, y = i / W;
    float mx = -INFINITY;
    for (int d = -p; d <= p; ++d) {
      const int yy = y + d;
      if (yy >= 0 && yy < H) mx = fmaxf(mx, s[yy * W + x]);
    }
    const float v = m[i];
    m[i] = (mx == v) ? v : v * 0.f;
  }
}

// ------------------------------------------------------------------ PCK (single block)
__global__ void __launch_bounds__(256) k_pck(const float* __restrict__ pred, const float* __restrict__ gt,
                                             const uint8_t* __restrict__ mask, const float* __restrict__ normalize,
                                             float thr, float* __restrict__ acc, float* __restrict__ avg_cnt, int N, int K) {
  __shared__ float s_acc[1024];
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    int valid = 0, hit = 0;
    for (int n = 0; n < N; ++n) {
      float n0 = normalize[n * 2], n1 = normalize[n * 2 + 1];
      bool mk = mask[n * K + k] != 0;
      if (n0 == 0.f || n1 == 0.f) mk = false;
      if (!mk) continue;
      if (n0 <= 0.f) n0 = 1e6f;
      if (n1 <= 0.f) n1 = 1e6f;
      const float dx = (pred[(n * K + k) * 2] - gt[(n * K + k) * 2]) / n0;
      const float dy = (pred[(n * K + k) * 2 + 1] - gt[(n * K + k) * 2 + 1]) / n1;
      const float d = sqrtf(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)));
      ++valid;
      hit += d < thr;
    }
    const float a = valid > 0 ? (float)((double)hit / (double)valid) : -1.f;
    acc[k] = a;
    s_acc[k] = a;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0;
    int c = 0;
    for (int k = 0; k < K; ++k)
      if (s_acc[k] >= 0.f) {
        s += (double)s_acc[k];
        ++c;
      }
    avg_cnt[0] = c > 0 ? (float)(s / c) : 0.f;
    avg_cnt[1] = (float)c;
  }
}

// ------------------------------------------------------------------ balanced MSE
// l = (o-t)^2 * w[n,k];  pos = t > 0.5;  loss = lw * (pf * S_pos + nf * S_neg) / numel
// pf = numel/(n_pos+1)*0.1, nf = numel/(n_neg+1)   (heatmapLoss.py:252-259)
__global__ void __launch_bounds__(256) k_loss_fwd(const float* __restrict__ o, const float* __restrict__ t,
                                                  const float* __restrict__ w, double* __restrict__ acc, int64_t NK, int HW) {
  double sp = 0, sn = 0;
  unsigned long long np = 0;
  const int64_t total4 = NK * HW / 4;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total4; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = q * 4;
    const float wk = w[e / HW];
    const f4 a = *reinterpret_cast<const f4*>(o + e), b = *reinterpret_cast<const f4*>(t + e);
    const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float d = av[j] - bv[j];
      const float l = mul_rn(mul_rn(d, d), wk);
      if (bv[j] > 0.5f) {
        sp += (double)l;
        ++np;
      } else {
        sn += (double)l;
      }
    }
  }
  sp = lhn_wave_sum_d(sp);
  sn = lhn_wave_sum_d(sn);
  double npd = lhn_wave_sum_d((double)np);
  __shared__ double red[4][3];
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[wv][0] = sp;
    red[wv][1] = sn;
    red[wv][2] = npd;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const double v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    atomicAdd(acc + 4 + (blockIdx.x & 15) * 4 + threadIdx.x, v);   // 16 replica slots after the 4 result doubles
  }
}
__global__ void k_loss_final(double* __restrict__ acc, float* __restrict__ loss, double numel, float lw, int balance) {
  double t0 = 0, t1 = 0, t2 = 0;
  for (int r = 0; r < 16; ++r) {
    t0 += acc[4 + r * 4 + 0];
    t1 += acc[4 + r * 4 + 1];
    t2 += acc[4 + r * 4 + 2];
  }
  acc[0] = t0;
  acc[1] = t1;
  acc[2] = t2;
  const double npos = acc[2], nneg = numel - npos;
  // the reference multiplies float32 loss elements by float32 factors; factors rounded to f32 here too
  const float pf = balance ? mul_rn((float)numel / (float)(npos + 1.0), 0.1f) : 1.f;
  const float nf = balance ? (float)numel / (float)(nneg + 1.0) : 1.f;
  loss[0] = lw * (float)(((double)pf * acc[0] + (double)nf * acc[1]) / numel);
}
__global__ void __launch_bounds__(256) k_loss_bwd(const float* __restrict__ o, const float* __restrict__ t,
                                                  const float* __restrict__ w, const double* __restrict__ acc,
                                                  const float* __restrict__ dloss, float* __restrict__ g, int64_t NK, int HW,
                                                  float lw, int balance) {
  const double numel = (double)NK * HW;
  const double npos = acc[2], nneg = numel - npos;
  const float pf = balance ? mul_rn((float)numel / (float)(npos + 1.0), 0.1f) : 1.f;
  const float nf = balance ? (float)numel / (float)(nneg + 1.0) : 1.f;
  const float up = (dloss ? dloss[0] : 1.f) * lw;
  const float kp = (float)((double)up * 2.0 * (double)pf / numel), kn = (float)((double)up * 2.0 * (double)nf / numel);
  const int64_t total4 = NK * HW / 4;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total4; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = q * 4;
    const float wk = w[e / HW];
    const f4 a = *reinterpret_cast<const f4*>(o + e), b = *reinterpret_cast<const f4*>(t + e);
    f4 r;
    r.x = (a.x - b.x) * wk * (b.x > 0.5f ? kp : kn);
    r.y = (a.y - b.y) * wk * (b.y > 0.5f ? kp : kn);
    r.z = (a.z - b.z) * wk * (b.z > 0.5f ? kp : kn);
    r.w = (a.w - b.w) * wk * (b.w > 0.5f ? kp : kn);
    *reinterpret_cast<f4*>(g + e) = r;
  }
}

// ------------------------------------------------------------------ C ABI
extern "C" {

int lhn_heatmap_encode(const float* joints, const float* visible, float* target, float* weight, int N, int K, int H,
                       int W, float img_w, float img_h, float sigma, int unbiased, void* stream) {
  LHN_CHECK_ARG(joints && visible && target && weight, "lhn_heatmap_encode: null pointer");
  LHN_CHECK_ARG(N > 0 && K > 0 && H > 0 && W > 0 && (H * W) % 4 == 0, "lhn_heatmap_encode: bad shape N=%d K=%d H=%d W=%d", N, K, H, W);
  LHN_CHECK_ARG(sigma > 0.f, "lhn_heatmap_encode: sigma must be > 0");
  LHN_CHECK_ARG(unbiased >= 0 && unbiased <= 2 && (unbiased != 2 || (W > 1 && H > 1)), "lhn_heatmap_encode: mode %d", unbiased);
  const double sx = unbiased == 2 ? ((double)img_w - 1.0) / ((double)W - 1.0) : (double)img_w / (double)W;
  const double sy = unbiased == 2 ? ((double)img_h - 1.0) / ((double)H - 1.0) : (double)img_h / (double)H;
  hipLaunchKernelGGL(k_encode, dim3(N * K), dim3(256), 0, (hipStream_t)stream, joints, visible, target, weight, N * K,
                     H, W, sx, sy, sigma, unbiased);
  LHN_CHECK_LAUNCH("lhn_heatmap_encode");
  return 0;
}

int lhn_heatmap_argmax(const float* hm, float* preds, float* maxvals, int32_t* index, int N, int K, int H, int W,
                       void* stream) {
  LHN_CHECK_ARG(hm && preds && maxvals, "lhn_heatmap_argmax: null pointer");
  LHN_CHECK_ARG(N > 0 && K > 0 && H > 0 && W > 0 && (H * W) % 4 == 0, "lhn_heatmap_argmax: bad shape");
  hipLaunchKernelGGL(k_decode, dim3(N * K), dim3(256), 0, (hipStream_t)stream, hm, (const float*)nullptr,
                     (const float*)nullptr, preds, (float*)nullptr, maxvals, index, K, H, W, 0);
  LHN_CHECK_LAUNCH("lhn_heatmap_argmax");
  return 0;
}

int lhn_heatmap_refine(const float* hm, float* preds, int N, int K, int H, int W, int mode, void* stream) {
  LHN_CHECK_ARG(hm && preds, "lhn_heatmap_refine: null pointer");
  LHN_CHECK_ARG(mode == 0 || mode == 1, "lhn_heatmap_refine: mode %d", mode);
  const int NK = N * K;
  hipLaunchKernelGGL(k_refine, dim3((NK + 255) / 256), dim3(256), 0, (hipStream_t)stream, hm, preds, NK, H, W, mode);
  LHN_CHECK_LAUNCH("lhn_heatmap_refine");
  return 0;
}

int lhn_transform_preds(const float* coords, const float* center, const float* scale, float* out, int N, int K, int W,
                        int H, int use_udp, void* stream) {
  LHN_CHECK_ARG(coords && center && scale && out, "lhn_transform_preds: null pointer");
  const int NK = N * K;
  hipLaunchKernelGGL(k_transform, dim3((NK + 255) / 256), dim3(256), 0, (hipStream_t)stream, coords, center, scale, out,
                     NK, K, W, H, use_udp);
  LHN_CHECK_LAUNCH("lhn_transform_preds");
  return 0;
}

int lhn_heatmap_decode(const float* hm, const float* center, const float* scale, float* hm_preds, float* preds,
                       float* maxvals, int N, int K, int H, int W, int post_process, void* stream) {
  LHN_CHECK_ARG(hm && center && scale && hm_preds && preds && maxvals, "lhn_heatmap_decode: null pointer");
  LHN_CHECK_ARG(post_process == 0 || post_process == 1, "lhn_heatmap_decode: post_process %d (0 none, 1 default)", post_process);
  LHN_CHECK_ARG((H * W) % 4 == 0, "lhn_heatmap_decode: H*W must be a multiple of 4");
  hipLaunchKernelGGL(k_decode, dim3(N * K), dim3(256), 0, (hipStream_t)stream, hm, center, scale, hm_preds, preds,
                     maxvals, (int32_t*)nullptr, K, H, W, post_process);
  LHN_CHECK_LAUNCH("lhn_heatmap_decode");
  return 0;
}

int lhn_heatmap_nms(float* hm, float* scratch, int N, int K, int H, int W, int kernel, void* stream) {
  LHN_CHECK_ARG(hm && scratch, "lhn_heatmap_nms: null pointer");
  LHN_CHECK_ARG(kernel % 2 == 1 && kernel > 0, "lhn_heatmap_nms: kernel must be odd");
  hipLaunchKernelGGL(k_nms, dim3(N * K), dim3(256), 0, (hipStream_t)stream, hm, scratch, H, W, kernel);
  LHN_CHECK_LAUNCH("lhn_heatmap_nms");
  return 0;
}

int lhn_pck_accuracy(const float* pred, const float* gt, const uint8_t* mask, const float* normalize, float thr,
                     float* acc, float* avg_cnt, int N, int K, void* stream) {
  LHN_CHECK_ARG(pred && gt && mask && normalize && acc && avg_cnt, "lhn_pck_accuracy: null pointer");
  LHN_CHECK_ARG(K <= 1024, "lhn_pck_accuracy: K <= 1024");
  hipLaunchKernelGGL(k_pck, dim3(1), dim3(256), 0, (hipStream_t)stream, pred, gt, mask, normalize, thr, acc, avg_cnt, N, K);
  LHN_CHECK_LAUNCH("lhn_pck_accuracy");
  return 0;
}

int lhn_loss_balanced_mse_fwd(const float* out, const float* target, const float* weight, double* acc, float* loss,
                              int64_t NK, int64_t HW, float loss_weight, int balance, void* stream) {
  LHN_CHECK_ARG(out && target && weight && acc && loss, "lhn_loss_balanced_mse_fwd: null pointer");
  LHN_CHECK_ARG(NK > 0 && HW > 0 && HW % 4 == 0, "lhn_loss_balanced_mse_fwd: bad shape");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(acc, 0, 68 * sizeof(double), s) != hipSuccess) {
    lhn_set_error("lhn_loss_balanced_mse_fwd: memset failed");
    return 2;
  }
  const int64_t total4 = NK * HW / 4;
  const int grid = (int)((total4 + 255) / 256 < (int64_t)lhn_num_cus() * 8 ? (total4 + 255) / 256 : (int64_t)lhn_num_cus() * 8);
  hipLaunchKernelGGL(k_loss_fwd, dim3(grid), dim3(256), 0, s, out, target, weight, acc, NK, (int)HW);
  hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(1), 0, s, acc, loss, (double)NK * (double)HW, loss_weight, balance);
  LHN_CHECK_LAUNCH("lhn_loss_balanced_mse_fwd");
  return 0;
}

int lhn_loss_balanced_mse_bwd(const float* out, const float* target, const float* weight, const double* acc,
                              const float* dloss, float* dout, int64_t NK, int64_t HW, float loss_weight, int balance,
                              void* stream) {
  LHN_CHECK_ARG(out && target && weight && acc && dout, "lhn_loss_balanced_mse_bwd: null pointer");
  LHN_CHECK_ARG(NK > 0 && HW > 0 && HW % 4 == 0, "lhn_loss_balanced_mse_bwd: bad shape");
  const int64_t total4 = NK * HW / 4;
  const int grid = (int)((total4 + 255) / 256 < (int64_t)lhn_num_cus() * 8 ? (total4 + 255) / 256 : (int64_t)lhn_num_cus() * 8);
  hipLaunchKernelGGL(k_loss_bwd, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, target, weight, acc, dloss, dout,
                     NK, (int)HW, loss_weight, balance);
  LHN_CHECK_LAUNCH("lhn_loss_balanced_mse_bwd");
  return 0;
}
}  // extern "C"

// ------------------------------------------------------------------ DARK ("unbiased") decode
// top_down_eval.py:233-272 (_gaussian_blur: k x k Gaussian on the zero-padded map, re-normalised to the original
// max), :433-439 (log(max(., 1e-10))), :338-372 (_taylor), then transform_preds.  One workgroup per (n, k) map,
// the whole map in LDS.  cv2.GaussianBlur is absent from the build container: the kernel follows OpenCV's
// documented rule sigma = 0.3*((k-1)*0.5-1)+0.8, normalised taps, separable, float32 -- "parity unpinned" at bit
// level, tolerance-tested against the numpy oracle.
__global__ void __launch_bounds__(256) k_dark(const float* __restrict__ hm, const float* __restrict__ center,
                                              const float* __restrict__ scale, float* __restrict__ hm_preds,
                                              float* __restrict__ preds, float* __restrict__ maxvals, int K, int H, int W,
                                              int ksize) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int HW = H * W, nk = blockIdx.x, b = (ksize - 1) / 2;
  float* A = smem;            // original map, later the blurred map
  float* B = smem + HW;       // row-blurred
  __shared__ float taps[32];
  __shared__ float sred[4];
  const float* m = hm + (int64_t)nk * HW;
  if (threadIdx.x < ksize) {
    const double sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8;
    double sum = 0;
    for (int t = 0; t < ksize; ++t) {
      const double xx = t - (ksize - 1) * 0.5;
      sum += exp(-(xx * xx) / (2 * sigma * sigma));
    }
    const double xx = threadIdx.x - (ksize - 1) * 0.5;
    taps[threadIdx.x] = (float)(exp(-(xx * xx) / (2 * sigma * sigma)) / sum);
  }
  for (int i = threadIdx.x; i < HW; i += blockDim.x) A[i] = m[i];
  const MaxI r = block_argmax(m, HW);          // contains a barrier
  __syncthreads();
  for (int i = threadIdx.x; i < HW; i += blockDim.x) {
    const int x = i % W, y = i / W;
    float s = 0.f;
    for (int t = 0; t < ksize; ++t) {
      const int xx = x + t - b;
      if (xx >= 0 && xx < W) s += taps[t] * A[y * W + xx];
    }
    B[i] = s;
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < HW; i += blockDim.x) {
    const int x = i % W, y = i / W;
    float s = 0.f;
    for (int t = 0; t < ksize; ++t) {
      const int yy = y + t - b;
      if (yy >= 0 && yy < H) s += taps[t] * B[yy * W + x];
    }
    A[i] = s;
    mx = fmaxf(mx, s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float bmax = fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3]));
    const float sc = r.v / bmax;               // heatmaps[i, j] *= origin_max / np.max(heatmaps[i, j])
    float x = (float)(r.i % W), y = (float)(r.i / W);
    if (!(r.v > 0.f)) x = y = -1.f;
    const int px = (int)x, py = (int)y;
    if (1 < px && px < W - 2 && 1 < py && py < H - 2) {
      auto L = [&](int yy, int xx) { return logf(fmaxf(A[yy * W + xx] * sc, 1e-10f)); };
      const float dx = 0.5f * (L(py, px + 1) - L(py, px - 1));
      const float dy = 0.5f * (L(py + 1, px) - L(py - 1, px));
      const float dxx = 0.25f * (L(py, px + 2) - 2.f * L(py, px) + L(py, px - 2));
      const float dxy = 0.25f * (L(py + 1, px + 1) - L(py - 1, px + 1) - L(py + 1, px - 1) + L(py - 1, px - 1));
      const float dyy = 0.25f * (L(py + 2, px) - 2.f * L(py, px) + L(py - 2, px));
      const float det = dxx * dyy - dxy * dxy;
      if (det != 0.f) {
        // offset = -H^-1 g
        x += -(dyy * dx - dxy * dy) / det;
        y += -(-dxy * dx + dxx * dy) / det;
      }
    }
    maxvals[nk] = r.v;
    hm_preds[nk * 2 + 0] = x;
    hm_preds[nk * 2 + 1] = y;
    const int n = nk / K;
    float ox, oy;
    xform_xy(x, y, center + n * 2, scale + n * 2, W, H, 0, ox, oy);
    preds[nk * 2 + 0] = ox;
    preds[nk * 2 + 1] = oy;
  }
}

extern "C" int lhn_heatmap_decode_dark(const float* hm, const float* center, const float* scale, float* hm_preds, float* preds,
                                       float* maxvals, int N, int K, int H, int W, int kernel, void* stream) {
  LHN_CHECK_ARG(hm && center && scale && hm_preds && preds && maxvals, "lhn_heatmap_decode_dark: null pointer");
  LHN_CHECK_ARG(kernel % 2 == 1 && kernel >= 3 && kernel <= 31, "lhn_heatmap_decode_dark: kernel %d (odd, 3..31)", kernel);
  LHN_CHECK_ARG((H * W) % 4 == 0 && H * W <= 16384, "lhn_heatmap_decode_dark: H*W must be a multiple of 4 and <= 16384");
  const size_t lds = (size_t)2 * H * W * sizeof(float);
  static LhnKernelCfg cfg;
  (void)lhn_kernel_cfg(cfg, &k_dark, (size_t)2 * 16384 * 4, 4, nullptr);
  hipLaunchKernelGGL(k_dark, dim3(N * K), dim3(256), lds, (hipStream_t)stream, hm, center, scale, hm_preds, preds, maxvals, K, H,
                     W, kernel);
  LHN_CHECK_LAUNCH("lhn_heatmap_decode_dark");
  return 0;
}

// ------------------------------------------------------------------ SimDR (1-D coordinate classification vectors)
// targets: datasets/data_pipeline/generate_simder.py:9-31; loss: loss/centernet_simdr_loss.py:6-71 (KLDiscretLoss with
// SmoothL1 'mean' + the per-joint weight quirk); decode reuses lhn_heatmap_argmax / lhn_transform_preds.
__global__ void __launch_bounds__(256) k_simdr_encode(const float* __restrict__ joints, const float* __restrict__ vis, int vis_stride,
                                                      float* __restrict__ tx, float* __restrict__ ty, int Wd, int Hd, float k,
                                                      float sigma) {
  const int nk = blockIdx.x;
  const bool on = vis[(size_t)nk * vis_stride] > 0.f;
  const float mux = mul_rn(joints[(size_t)nk * 3 + 0], k), muy = mul_rn(joints[(size_t)nk * 3 + 1], k);
  const float den = 2.f * sigma * sigma;
  for (int i = threadIdx.x; i < Wd; i += blockDim.x) {
    const float d = sub_rn((float)i, mux);
    tx[(size_t)nk * Wd + i] = on ? expf(-mul_rn(d, d) / den) : 0.f;
  }
  for (int i = threadIdx.x; i < Hd; i += blockDim.x) {
    const float d = sub_rn((float)i, muy);
    ty[(size_t)nk * Hd + i] = on ? expf(-mul_rn(d, d) / den) : 0.f;
  }
}

__device__ __forceinline__ float smooth_l1(float d) {
  const float a = fabsf(d);
  return a < 1.f ? 0.5f * d * d : a - 0.5f;
}
// one block per joint index: sums[idx] = sum_{b,i} smoothl1(px - tx), sums[K + idx] = same for y, sums[2K + idx] = sum_b w
__global__ void __launch_bounds__(256) k_simdr_loss_sums(const float* __restrict__ px, const float* __restrict__ py,
                                                         const float* __restrict__ tx, const float* __restrict__ ty,
                                                         const float* __restrict__ w, double* __restrict__ sums, int N, int K,
                                                         int Wd, int Hd) {
  __shared__ double rs[256], rq[256];
  const int idx = blockIdx.x;
  double sx = 0, sy = 0;
  for (int b = 0; b < N; ++b) {
    const size_t ox = ((size_t)b * K + idx) * Wd, oy = ((size_t)b * K + idx) * Hd;
    for (int i = threadIdx.x; i < Wd; i += blockDim.x) sx += smooth_l1(px[ox + i] - tx[ox + i]);
    for (int i = threadIdx.x; i < Hd; i += blockDim.x) sy += smooth_l1(py[oy + i] - ty[oy + i]);
  }
  rs[threadIdx.x] = sx;
  rq[threadIdx.x] = sy;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      rs[threadIdx.x] += rs[threadIdx.x + o];
      rq[threadIdx.x] += rq[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double sw = 0;
    for (int b = 0; b < N; ++b) sw += w[(size_t)b * K + idx];
    sums[idx] = rs[0];
    sums[K + idx] = rq[0];
    sums[2 * K + idx] = sw;
  }
}
__global__ void k_simdr_loss_final(const double* __restrict__ sums, float* __restrict__ loss, int N, int K, int Wd, int Hd) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double l = 0;
    for (int idx = 0; idx < K; ++idx) {
      const double mw = sums[2 * K + idx] / N;
      l += (sums[idx] / ((double)N * Wd) + sums[K + idx] / ((double)N * Hd)) * mw;
    }
    loss[0] = (float)(l / K);
  }
}
// d loss / d px = gout * mean_w[idx] / (K * N * Wd) * smoothl1'(px - tx)
__global__ void __launch_bounds__(256) k_simdr_loss_bwd(const float* __restrict__ p, const float* __restrict__ t,
                                                        const double* __restrict__ sums, const float* __restrict__ gout,
                                                        float* __restrict__ dp, int N, int K, int L) {
  const size_t total = (size_t)N * K * L;
  const float g = gout[0];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int idx = (int)((i / L) % K);
    const float c = g * (float)(sums[2 * K + idx] / N / ((double)K * N * L));
    const float d = p[i] - t[i];
    dp[i] = c * (fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : -1.f));
  }
}

extern "C" int lhn_simdr_encode(const float* joints, const float* visible, int vis_stride, float* target_x, float* target_y, int N,
                                int K, int Wd, int Hd, float k, float sigma, void* stream) {
  LHN_CHECK_ARG(joints && visible && target_x && target_y && N > 0 && K > 0 && Wd > 0 && Hd > 0 && k > 0 && sigma > 0,
                "lhn_simdr_encode: bad argument");
  hipLaunchKernelGGL(k_simdr_encode, dim3(N * K), dim3(256), 0, (hipStream_t)stream, joints, visible, vis_stride, target_x, target_y,
                     Wd, Hd, k, sigma);
  LHN_CHECK_LAUNCH("lhn_simdr_encode");
  return 0;
}
extern "C" int lhn_simdr_loss_fwd(const float* px, const float* py, const float* tx, const float* ty, const float* weight,
                                  double* sums, float* loss, int N, int K, int Wd, int Hd, void* stream) {
  LHN_CHECK_ARG(px && py && tx && ty && weight && sums && loss && N > 0 && K > 0 && Wd > 0 && Hd > 0, "lhn_simdr_loss_fwd: bad argument");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_simdr_loss_sums, dim3(K), dim3(256), 0, s, px, py, tx, ty, weight, sums, N, K, Wd, Hd);
  hipLaunchKernelGGL(k_simdr_loss_final, dim3(1), dim3(64), 0, s, sums, loss, N, K, Wd, Hd);
  LHN_CHECK_LAUNCH("lhn_simdr_loss_fwd");
  return 0;
}
extern "C" int lhn_simdr_loss_bwd(const float* px, const float* py, const float* tx, const float* ty, const double* sums,
                                  const float* gout, float* dpx, float* dpy, int N, int K, int Wd, int Hd, void* stream) {
  LHN_CHECK_ARG(px && py && tx && ty && sums && gout && dpx && dpy, "lhn_simdr_loss_bwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_simdr_loss_bwd, dim3(512), dim3(256), 0, s, px, tx, sums, gout, dpx, N, K, Wd);
  hipLaunchKernelGGL(k_simdr_loss_bwd, dim3(512), dim3(256), 0, s, py, ty, sums, gout, dpy, N, K, Hd);
  LHN_CHECK_LAUNCH("lhn_simdr_loss_bwd");
  return 0;
}

// ------------------------------------------------------------------ GPU input path: TopDownAffine + ToTensor + Normalize
// datasets/data_pipeline/topdown_affine.py:47-114 (non-UDP branch) + shared_transform.py:3-44, fused:
//   trans = get_affine_transform(center, scale, rot, out_size)  (post_transforms.py:101-156: a similarity transform
//           dst = s * R(-rot) * (src - center) + out_size/2 with s = out_w / (scale_x * 200))
//   img   = warpAffine(img_u8, trans, INTER_LINEAR, constant 0 border) -> /255 -> (v - mean) / std, CHW
//   joints[:, :2] = trans * [x, y, 1] for visible joints
// cv2 is absent from the build container: the bilinear sample is exact float arithmetic rounded to the nearest uint8
// level like cv2's output type (cv2 itself interpolates with 5-bit fixed-point weights) -- "parity unpinned".
__global__ void __launch_bounds__(256) k_affine_warp_norm(const unsigned char* __restrict__ img, int Hs, int Ws,
                                                          const float* __restrict__ center, const float* __restrict__ scale,
                                                          const float* __restrict__ rot, float m0, float m1, float m2, float s0,
                                                          float s1, float s2, float* __restrict__ out, int Ho, int Wo, int udp,
                                                          const unsigned char* __restrict__ flipped) {
  const int n = blockIdx.y;
  const bool flip = flipped && flipped[n];      // TopDownRandomFlip (RandomFlip.py:40-41): the source image is read mirrored
  const float r = rot[n] * 3.14159265358979323846f / 180.f;
  const float cr = cosf(r), sr = sinf(r);
  // src = (qx, qy) + [[a00, a01], [a10, a11]] * (dst - (px, py))
  float a00, a01, a10, a11, qx, qy, px, py;
  if (!udp) {   // get_affine_transform: dst = s R(-rot) (src - center) + out/2
    const float inv_s = scale[n * 2] * 200.f / (float)Wo;
    a00 = inv_s * cr; a01 = -inv_s * sr; a10 = inv_s * sr; a11 = inv_s * cr;
    qx = center[n * 2]; qy = center[n * 2 + 1]; px = 0.5f * Wo; py = 0.5f * Ho;
  } else {      // get_warp_matrix(rot, center*2, image_size-1, scale*200) (post_transforms.py:49-83): dst = S R(rot) src + t
    const float tw = scale[n * 2] * 200.f, th = scale[n * 2 + 1] * 200.f, iw = center[n * 2] * 2.f, ih = center[n * 2 + 1] * 2.f;
    const float sx = ((float)Wo - 1.f) / tw, sy = ((float)Ho - 1.f) / th;
    a00 = cr / sx; a01 = sr / sy; a10 = -sr / sx; a11 = cr / sy;
    qx = 0.f; qy = 0.f;
    px = sx * (-0.5f * iw * cr + 0.5f * ih * sr + 0.5f * tw);
    py = sy * (-0.5f * iw * sr - 0.5f * ih * cr + 0.5f * th);
  }
  const unsigned char* src = img + (size_t)n * Hs * Ws * 3;
  const float mean[3] = {m0, m1, m2}, istd[3] = {1.f / s0, 1.f / s1, 1.f / s2};
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < Ho * Wo; p += gridDim.x * blockDim.x) {
    const int y = p / Wo, x = p - y * Wo;
    const float dx = (float)x - px, dy = (float)y - py;
    const float sx = qx + a00 * dx + a01 * dy, sy = qy + a10 * dx + a11 * dy;
    const float fx = floorf(sx), fy = floorf(sy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float ax = sx - fx, ay = sy - fy;
    float v[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int xx = x0 + i, yy = y0 + j;
        if (xx >= 0 && xx < Ws && yy >= 0 && yy < Hs) {
          const float wgt = (i ? ax : 1.f - ax) * (j ? ay : 1.f - ay);
          const unsigned char* q = src + ((size_t)yy * Ws + (flip ? Ws - 1 - xx : xx)) * 3;
          v[0] += wgt * q[0];
          v[1] += wgt * q[1];
          v[2] += wgt * q[2];
        }
      }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float u8 = fminf(fmaxf(rintf(v[c]), 0.f), 255.f);     // cv2.warpAffine returns uint8
      out[((size_t)n * 3 + c) * Ho * Wo + p] = (u8 / 255.f - mean[c]) * istd[c];
    }
  }
}
__global__ void k_affine_joints(float* __restrict__ joints, const float* __restrict__ visible, int vis_stride,
                                const float* __restrict__ center, const float* __restrict__ scale, const float* __restrict__ rot,
                                int K, int Ho, int Wo, int total, int udp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int n = i / K;
  const float r = rot[n] * 3.14159265358979323846f / 180.f, cr = cosf(r), sr = sinf(r);
  if (udp) {   // warp_affine_joints (post_transforms.py:86-100): every joint, visible or not
    const float tw = scale[n * 2] * 200.f, th = scale[n * 2 + 1] * 200.f, iw = center[n * 2] * 2.f, ih = center[n * 2 + 1] * 2.f;
    const float sx = ((float)Wo - 1.f) / tw, sy = ((float)Ho - 1.f) / th;
    const float x = joints[(size_t)i * 3], y = joints[(size_t)i * 3 + 1];
    joints[(size_t)i * 3] = cr * sx * x - sr * sx * y + sx * (-0.5f * iw * cr + 0.5f * ih * sr + 0.5f * tw);
    joints[(size_t)i * 3 + 1] = sr * sy * x + cr * sy * y + sy * (-0.5f * iw * sr - 0.5f * ih * cr + 0.5f * th);
    return;
  }
  if (!(visible[(size_t)i * vis_stride] > 0.f)) return;
  const float s = (float)Wo / (scale[n * 2] * 200.f);
  const float dx = joints[(size_t)i * 3] - center[n * 2], dy = joints[(size_t)i * 3 + 1] - center[n * 2 + 1];
  joints[(size_t)i * 3] = s * (cr * dx + sr * dy) + 0.5f * Wo;            // dst = s * R(-rot) * (src - center) + out/2
  joints[(size_t)i * 3 + 1] = s * (-sr * dx + cr * dy) + 0.5f * Ho;
}

// HeatmapParser.candidate_bbox (utils/HeatmapParser.py:52-85): the k highest values of every (NMS-suppressed) centre map in
// descending order -> (x, y) = (idx % W, idx // W) * feature_stride, confidence, and the box width / height ratios sampled
// from the (already region-averaged) size maps at the peak, clipped to [0, 0.99] and scaled by the image size.
// One block per map: the map sits in LDS, k rounds of block arg-max (first index wins ties, like a stable sort) each
// knocking its winner out.  k is small (num_candidates ~ 10-30), H*W <= 16384.
__global__ void __launch_bounds__(256) k_topk_candidates(const float* __restrict__ centre, const float* __restrict__ sizes,
                                                         float* __restrict__ cand, int H, int W, int k, float stride,
                                                         float image_size) {
  extern __shared__ __attribute__((aligned(16))) float A[];
  const int n = blockIdx.x, HW = H * W;
  const float* m = centre + (size_t)n * HW;
  for (int i = threadIdx.x; i < HW; i += blockDim.x) A[i] = m[i];
  __syncthreads();
  for (int t = 0; t < k; ++t) {
    const MaxI r = block_argmax(A, HW);          // contains barriers; every thread gets the winner
    __syncthreads();
    if (threadIdx.x == 0) {
      float* c = cand + ((size_t)n * k + t) * 5;
      const int x = r.i % W, y = r.i / W;
      c[0] = (float)x * stride;
      c[1] = (float)y * stride;
      float bw = 0.f, bh = 0.f;
      if (sizes) {
        bw = fminf(fmaxf(sizes[((size_t)n * 2 + 0) * HW + r.i], 0.f), 0.99f);
        bh = fminf(fmaxf(sizes[((size_t)n * 2 + 1) * HW + r.i], 0.f), 0.99f);
      }
      c[2] = bw * image_size;
      c[3] = bh * image_size;
      c[4] = r.v;
      A[r.i] = -INFINITY;
    }
    __syncthreads();
  }
}

extern "C" int lhn_heatmap_topk(const float* centre_maps, const float* size_maps, float* candidates, int N, int H, int W, int k,
                                float image_size, void* stream) {
  LHN_CHECK_ARG(centre_maps && candidates && N > 0 && H > 0 && W > 0 && k > 0 && k <= H * W, "lhn_heatmap_topk: bad argument");
  LHN_CHECK_ARG((H * W) % 4 == 0 && H * W <= 16384, "lhn_heatmap_topk: H*W must be a multiple of 4 and <= 16384");
  static LhnKernelCfg cfg;
  (void)lhn_kernel_cfg(cfg, &k_topk_candidates, (size_t)16384 * 4, 4, nullptr);
  hipLaunchKernelGGL(k_topk_candidates, dim3(N), dim3(256), (size_t)H * W * 4, (hipStream_t)stream, centre_maps, size_maps, candidates, H,
                     W, k, image_size / (float)W, image_size);
  LHN_CHECK_LAUNCH("lhn_heatmap_topk");
  return 0;
}

// TopDownRandomFlip.__call__ + fliplr_joints (RandomFlip.py:28-100) for the samples flagged in `flipped`, minus the image
// (lhn_affine_warp_normalize2 reads it mirrored): joints / visibility of every (left, right) pair exchanged FROM THE ORIGINAL
// arrays, x -> W - 1 - x for every joint, joints *= visibility, center_x -> W - 1 - center_x.  One block per sample.
__global__ void __launch_bounds__(64) k_flip_joints(float* __restrict__ joints, float* __restrict__ visible, int vs,
                                                    float* __restrict__ center, const unsigned char* __restrict__ flipped,
                                                    const int* __restrict__ pairs, int npairs, int K, float img_w) {
  __shared__ float sj[64 * 3], sv[64 * 3];
  __shared__ int src[64];
  const int n = blockIdx.x, k = threadIdx.x;
  if (!flipped[n]) return;
  if (k < K) {
    src[k] = k;
    for (int c = 0; c < 3; ++c) {
      sj[k * 3 + c] = joints[((size_t)n * K + k) * 3 + c];
      sv[k * 3 + c] = visible[((size_t)n * K + k) * vs + (c < vs ? c : vs - 1)];
    }
  }
  __syncthreads();
  if (k == 0)
    for (int p = 0; p < npairs; ++p) {           // later pairs win, as the reference's sequential assignments do
      const int l = pairs[2 * p], r = pairs[2 * p + 1];
      if (l >= 0 && l < K && r >= 0 && r < K) {
        src[l] = r;
        src[r] = l;
      }
    }
  __syncthreads();
  if (k < K) {
    const int q = src[k];
    float v[3], jx[3];
    for (int c = 0; c < 3; ++c) {
      v[c] = sv[q * 3 + c];
      jx[c] = sj[q * 3 + c];
    }
    jx[0] = img_w - 1.f - jx[0];
    for (int c = 0; c < 3; ++c) {
      joints[((size_t)n * K + k) * 3 + c] = jx[c] * v[c];
      if (c < vs) visible[((size_t)n * K + k) * vs + c] = v[c];
    }
  }
  if (k == 0) center[n * 2] = img_w - center[n * 2] - 1.f;
}

// ------------------------------------------------------------------ HSVRandomAug (datasets/data_pipeline/random_hsv.py:20-34)
// One thread per pixel: 8-bit BGR -> HSV (OpenCV's fixed-point RGB2HSV_b, hue range 180), the reference's integer jitter
// ((h + dh) mod 180, clip(s + ds), clip(v + dv) in int16), HSV -> BGR (OpenCV's float sector formula, rounded to nearest even).
// cv2 is not available where this was written: the colour conversions follow OpenCV's published algorithm, parity unpinned.
__device__ __forceinline__ int lhn_cv_round_div(int num, double den) { return (int)__builtin_rint((double)num / den); }
__global__ void __launch_bounds__(256) k_hsv_jitter(unsigned char* __restrict__ img, const short* __restrict__ gains, int64_t hw, int N) {
  const int64_t total = hw * N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int n = (int)(i / hw);
    unsigned char* p = img + i * 3;
    const int b = p[0], g = p[1], r = p[2];
    const int v = max(max(b, g), r), vmin = min(min(b, g), r), diff = v - vmin;
    const int sdiv = v > 0 ? lhn_cv_round_div(255 << 12, (double)v) : 0;
    const int hdiv = diff > 0 ? lhn_cv_round_div(180 << 12, 6.0 * diff) : 0;
    int s = (diff * sdiv + (1 << 11)) >> 12;
    int h = v == r ? g - b : (v == g ? b - r + 2 * diff : r - g + 4 * diff);
    h = (h * hdiv + (1 << 11)) >> 12;
    if (h < 0) h += 180;
    h = min(max(h, 0), 255);
    // the reference's jitter on int16 planes
    const int dh = gains[n * 3 + 0], ds = gains[n * 3 + 1], dv = gains[n * 3 + 2];
    int hj = (h + dh) % 180;
    if (hj < 0) hj += 180;                                   // numpy's % is non-negative for a positive modulus
    const int sj = min(max(s + ds, 0), 255), vj = min(max(v + dv, 0), 255);
    // HSV -> BGR
    float hf = (float)hj * (float)(6.0 / 180.0);
    const float sf = (float)sj * (float)(1.0 / 255.0), vf = (float)vj * (float)(1.0 / 255.0);
    float bo, go, ro;
    if (sj == 0) {
      bo = go = ro = vf;
    } else {
      if (hf >= 6.f) hf -= 6.f;
      int sector = (int)floorf(hf);
      float f = hf - (float)sector;
      if (sector < 0 || sector >= 6) {
        sector = 0;
        f = 0.f;
      }
      const float t0 = vf, t1 = vf * (1.f - sf), t2 = vf * (1.f - sf * f), t3 = vf * (1.f - sf * (1.f - f));
      switch (sector) {
        case 0: bo = t1; go = t3; ro = t0; break;
        case 1: bo = t1; go = t0; ro = t2; break;
        case 2: bo = t3; go = t0; ro = t1; break;
        case 3: bo = t0; go = t2; ro = t1; break;
        case 4: bo = t0; go = t1; ro = t3; break;
        default: bo = t2; go = t1; ro = t0; break;
      }
    }
    p[0] = (unsigned char)min(max((int)rintf(bo * 255.f), 0), 255);
    p[1] = (unsigned char)min(max((int)rintf(go * 255.f), 0), 255);
    p[2] = (unsigned char)min(max((int)rintf(ro * 255.f), 0), 255);
  }
}
extern "C" int lhn_hsv_jitter(unsigned char* img, const int16_t* gains, int N, int H, int W, void* stream) {
  LHN_CHECK_ARG(img && gains && N > 0 && H > 0 && W > 0, "lhn_hsv_jitter: bad argument");
  const int64_t hw = (int64_t)H * W;
  int64_t blocks = (hw * N + 255) / 256;
  const int64_t cap = (int64_t)lhn_num_cus() * 16;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(k_hsv_jitter, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, img, gains, hw, N);
  LHN_CHECK_LAUNCH("lhn_hsv_jitter");
  return 0;
}

extern "C" int lhn_random_flip(float* joints, float* visible, int vis_stride, float* center, const unsigned char* flipped,
                               const int32_t* pairs, int npairs, int N, int K, int img_width, void* stream) {
  LHN_CHECK_ARG(joints && visible && center && flipped && (pairs || npairs == 0) && N > 0 && K > 0 && K <= 64 && npairs >= 0 &&
                    vis_stride >= 1 && vis_stride <= 3 && img_width > 0,
                "lhn_random_flip: bad argument (K <= 64, visibility columns 1..3)");
  hipLaunchKernelGGL(k_flip_joints, dim3(N), dim3(64), 0, (hipStream_t)stream, joints, visible, vis_stride, center, flipped, pairs,
                     npairs, K, (float)img_width);
  LHN_CHECK_LAUNCH("lhn_random_flip");
  return 0;
}

extern "C" int lhn_affine_warp_normalize2(const unsigned char* img, int N, int Hs, int Ws, const float* center, const float* scale,
                                          const float* rot, const float* mean3, const float* std3, float* out, int Ho, int Wo,
                                          float* joints, const float* visible, int vis_stride, int K, int use_udp,
                                          const unsigned char* flipped, void* stream);
extern "C" int lhn_affine_warp_normalize(const unsigned char* img, int N, int Hs, int Ws, const float* center, const float* scale,
                                         const float* rot, const float* mean3, const float* std3, float* out, int Ho, int Wo,
                                         float* joints, const float* visible, int vis_stride, int K, int use_udp, void* stream) {
  return lhn_affine_warp_normalize2(img, N, Hs, Ws, center, scale, rot, mean3, std3, out, Ho, Wo, joints, visible, vis_stride, K,
                                    use_udp, nullptr, stream);
}
extern "C" int lhn_affine_warp_normalize2(const unsigned char* img, int N, int Hs, int Ws, const float* center, const float* scale,
                                          const float* rot, const float* mean3, const float* std3, float* out, int Ho, int Wo,
                                          float* joints, const float* visible, int vis_stride, int K, int use_udp,
                                          const unsigned char* flipped, void* stream) {
  LHN_CHECK_ARG(img && center && scale && rot && mean3 && std3 && out && N > 0 && Hs > 0 && Ws > 0 && Ho > 0 && Wo > 0,
                "lhn_affine_warp_normalize: bad argument");
  LHN_CHECK_ARG(!joints || (visible && K > 0), "lhn_affine_warp_normalize: joints need visibility flags");
  hipStream_t s = (hipStream_t)stream;
  int gx = (Ho * Wo + 255) / 256;
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(k_affine_warp_norm, dim3(gx, N), dim3(256), 0, s, img, Hs, Ws, center, scale, rot, mean3[0], mean3[1], mean3[2],
                     std3[0], std3[1], std3[2], out, Ho, Wo, use_udp, flipped);
  if (joints)
    hipLaunchKernelGGL(k_affine_joints, dim3((N * K + 255) / 256), dim3(256), 0, s, joints, visible, vis_stride, center, scale, rot, K,
                       Ho, Wo, N * K, use_udp);
  LHN_CHECK_LAUNCH("lhn_affine_warp_normalize");
  return 0;
}


// ------------------------------------------------------------------ UDP + DARK decode (use_udp, 'unbiased')
// top_down_eval.py:275-337 (post_dark_udp) behind keypoints_from_heatmaps(use_udp=True, target_type GaussianHeatmap,
// :404-411): blur the map in place (cv2.GaussianBlur, default REFLECT_101 border -- parity unpinned, cv2 absent), clip to
// [0.001, 50], log, Newton step with edge-replicated finite differences and (Hessian + eps*I)^-1, then the UDP
// back-transform (post_transforms.py:37-43: scale / (size - 1)).
__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
  return i;
}
__global__ void __launch_bounds__(256) k_dark_udp(const float* __restrict__ hm, const float* __restrict__ center,
                                                  const float* __restrict__ scale, float* __restrict__ hm_preds,
                                                  float* __restrict__ preds, float* __restrict__ maxvals, int K, int H, int W,
                                                  int ksize) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int HW = H * W, nk = blockIdx.x, b = (ksize - 1) / 2;
  float* A = smem;
  float* B = smem + HW;
  __shared__ float taps[32];
  const float* m = hm + (int64_t)nk * HW;
  if (threadIdx.x < ksize) {
    const double sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8;
    double sum = 0;
    for (int t = 0; t < ksize; ++t) {
      const double xx = t - (ksize - 1) * 0.5;
      sum += exp(-(xx * xx) / (2 * sigma * sigma));
    }
    const double xx = threadIdx.x - (ksize - 1) * 0.5;
    taps[threadIdx.x] = (float)(exp(-(xx * xx) / (2 * sigma * sigma)) / sum);
  }
  for (int i = threadIdx.x; i < HW; i += blockDim.x) A[i] = m[i];
  const MaxI r = block_argmax(m, HW);
  __syncthreads();
  for (int i = threadIdx.x; i < HW; i += blockDim.x) {
    const int x = i % W, y = i / W;
    float s = 0.f;
    for (int t = 0; t < ksize; ++t) s += taps[t] * A[y * W + reflect101(x + t - b, W)];
    B[i] = s;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < HW; i += blockDim.x) {
    const int x = i % W, y = i / W;
    float s = 0.f;
    for (int t = 0; t < ksize; ++t) s += taps[t] * B[reflect101(y + t - b, H) * W + x];
    A[i] = logf(fminf(fmaxf(s, 0.001f), 50.f));
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float x = (float)(r.i % W), y = (float)(r.i / W);
    if (!(r.v > 0.f)) x = y = -1.f;
    // np.pad(..., mode='edge') + index arithmetic: coordinates are clamped into the padded map
    auto L = [&](int yy, int xx) { return A[min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1)]; };
    const int px = (int)x, py = (int)y;
    const float i_ = L(py, px), ix1 = L(py, px + 1), iy1 = L(py + 1, px), ix1y1 = L(py + 1, px + 1);
    const float ix1_y1_ = L(py - 1, px - 1), ix1_ = L(py, px - 1), iy1_ = L(py - 1, px);
    const float dx = 0.5f * (ix1 - ix1_), dy = 0.5f * (iy1 - iy1_);
    const float eps = 1.1920929e-07f;
    const float dxx = ix1 - 2.f * i_ + ix1_ + eps, dyy = iy1 - 2.f * i_ + iy1_ + eps;
    const float dxy = 0.5f * (ix1y1 - ix1 - iy1 + i_ + i_ - ix1_ - iy1_ + ix1_y1_);
    const float det = dxx * dyy - dxy * dxy;
    x -= (dyy * dx - dxy * dy) / det;
    y -= (-dxy * dx + dxx * dy) / det;
    maxvals[nk] = r.v;
    hm_preds[nk * 2 + 0] = x;
    hm_preds[nk * 2 + 1] = y;
    const int n = nk / K;
    float ox, oy;
    xform_xy(x, y, center + n * 2, scale + n * 2, W, H, 1, ox, oy);
    preds[nk * 2 + 0] = ox;
    preds[nk * 2 + 1] = oy;
  }
}

extern "C" int lhn_heatmap_decode_dark_udp(const float* hm, const float* center, const float* scale, float* hm_preds, float* preds,
                                           float* maxvals, int N, int K, int H, int W, int kernel, void* stream) {
  LHN_CHECK_ARG(hm && center && scale && hm_preds && preds && maxvals, "lhn_heatmap_decode_dark_udp: null pointer");
  LHN_CHECK_ARG(kernel % 2 == 1 && kernel >= 3 && kernel <= 31, "lhn_heatmap_decode_dark_udp: kernel %d (odd, 3..31)", kernel);
  LHN_CHECK_ARG((H * W) % 4 == 0 && H * W <= 16384 && H > 1 && W > 1, "lhn_heatmap_decode_dark_udp: H*W must be a multiple of 4 and <= 16384");
  const size_t lds = (size_t)2 * H * W * sizeof(float);
  static LhnKernelCfg cfg;
  (void)lhn_kernel_cfg(cfg, &k_dark_udp, (size_t)2 * 16384 * 4, 4, nullptr);
  hipLaunchKernelGGL(k_dark_udp, dim3(N * K), dim3(256), lds, (hipStream_t)stream, hm, center, scale, hm_preds, preds, maxvals, K,
                     H, W, kernel);
  LHN_CHECK_LAUNCH("lhn_heatmap_decode_dark_udp");
  return 0;
}
