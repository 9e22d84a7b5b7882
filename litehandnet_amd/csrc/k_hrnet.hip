// Elementwise pieces that only the Lite-HRNet baseline needs (models/pose_estimation/lite_hrnet.py; BASELINE config 5):
//   channel shuffle of a two-way concatenation (:29-52,141-142,246-247), the product with a nearest-upsampled weight map
//   (:105-107) and its backward, and the backward of the bilinear (align_corners=True) upsample-add (:272-275).
// NHWC fp32, thread = (group of 4 destination channels, pixel lane), HBM-bound; the forward of the product / bilinear
// combine lives in k_ew_fwd (k_misc.hip: EwSrcs.mode).
#include "lhn_common.h"

static inline int grid_cap(int64_t blocks, int cap_per_cu) {
  const int64_t cap = (int64_t)lhn_num_cus() * cap_per_cu;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

// consumed value of 2 consecutive channels (absolute index c, even) at a pixel
struct Xf2 {
  float sc[2], sh[2], sl[2];
};
__device__ __forceinline__ Xf2 load_xf2(const lhn_view& v, int c) {
  Xf2 t;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    t.sc[j] = v.table ? v.table[c + j] : 1.f;
    t.sh[j] = v.table ? v.table[v.cstride + c + j] : 0.f;
    t.sl[j] = v.table ? v.table[2 * v.cstride + c + j] : 1.f;
  }
  return t;
}

// ------------------------------------------------------------------ channel_shuffle(cat(a, b), groups = 2)
// dst channel 2j = a[j], 2j+1 = b[j]; dst is stored plain (table identity).  One thread writes 4 dst channels =
// a[2g], b[2g], a[2g+1], b[2g+1].
__global__ void __launch_bounds__(256) k_shuffle2_fwd(lhn_view a, lhn_view b, lhn_view dst) {
  const int C4 = dst.C >> 2, c4 = threadIdx.x % C4, pl = threadIdx.x / C4, PL = 256 / C4;
  const int ca = a.coff + 2 * c4, cb = b.coff + 2 * c4;
  const Xf2 xa = load_xf2(a, ca), xb = load_xf2(b, cb);
  const int rows = dst.N * dst.H;
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const int n = row / dst.H;
    float ga[2] = {1.f, 1.f}, gb[2] = {1.f, 1.f};
    if (a.gate) { ga[0] = a.gate[(size_t)n * a.cstride + ca]; ga[1] = a.gate[(size_t)n * a.cstride + ca + 1]; }
    if (b.gate) { gb[0] = b.gate[(size_t)n * b.cstride + cb]; gb[1] = b.gate[(size_t)n * b.cstride + cb + 1]; }
    for (int w = LHN_LANE0(pl, PL); w < dst.W; w += PL) {
      const size_t pix = (size_t)row * dst.W + w;
      const float2 ra = *reinterpret_cast<const float2*>(a.data + pix * a.cstride + ca);
      const float2 rb = *reinterpret_cast<const float2*>(b.data + pix * b.cstride + cb);
      f4 o;
      o.x = lhn_lrelu(ra.x * xa.sc[0] + xa.sh[0], xa.sl[0]) * ga[0];
      o.y = lhn_lrelu(rb.x * xb.sc[0] + xb.sh[0], xb.sl[0]) * gb[0];
      o.z = lhn_lrelu(ra.y * xa.sc[1] + xa.sh[1], xa.sl[1]) * ga[1];
      o.w = lhn_lrelu(rb.y * xb.sc[1] + xb.sh[1], xb.sl[1]) * gb[1];
      *reinterpret_cast<f4*>(dst.data + pix * dst.cstride + dst.coff + 4 * c4) = o;
    }
  }
}
// d(value of a)[j] (+)= d(dst)[2j], d(value of b)[j] (+)= d(dst)[2j+1]   (da / db: gradient buffers with a's / b's geometry)
__global__ void __launch_bounds__(256) k_shuffle2_bwd(lhn_view a, lhn_view b, lhn_view dst, const float* __restrict__ ddst,
                                                      float* __restrict__ da, int acc_a, float* __restrict__ db, int acc_b) {
  const int C4 = dst.C >> 2, c4 = threadIdx.x % C4, pl = threadIdx.x / C4, PL = 256 / C4;
  const int ca = a.coff + 2 * c4, cb = b.coff + 2 * c4;
  const int rows = dst.N * dst.H;
  for (int row = blockIdx.x; row < rows; row += gridDim.x)
    for (int w = LHN_LANE0(pl, PL); w < dst.W; w += PL) {
      const size_t pix = (size_t)row * dst.W + w;
      const f4 g = *reinterpret_cast<const f4*>(ddst + pix * dst.cstride + dst.coff + 4 * c4);
      if (da) {
        float2* o = reinterpret_cast<float2*>(da + pix * a.cstride + ca);
        float2 v = make_float2(g.x, g.z);
        if (acc_a) { v.x += o->x; v.y += o->y; }
        *o = v;
      }
      if (db) {
        float2* o = reinterpret_cast<float2*>(db + pix * b.cstride + cb);
        float2 v = make_float2(g.y, g.w);
        if (acc_b) { v.x += o->x; v.y += o->y; }
        *o = v;
      }
    }
}

// ------------------------------------------------------------------ backward of dst = value(a) * up_nearest(value(g))
// One launch per operand, gather form (every source element sums the destination elements that read it):
//   d(src)[p] (+)= sum_{q reads p} d(dst)[q] * value(other at q)
__device__ __forceinline__ int nearest_src2(int d, int in, int out) {
  if (in == out) return d;
  const float sc = (float)in / (float)out;
  const int s = (int)floorf((float)d * sc);
  return s < in - 1 ? s : in - 1;
}
__global__ void __launch_bounds__(256) k_ew_mul_bwd(lhn_view src, lhn_view other, lhn_view dst, const float* __restrict__ ddst,
                                                    float* __restrict__ dsrc, int accumulate) {
  const int C4 = src.C >> 2, c4 = threadIdx.x % C4, pl = threadIdx.x / C4, PL = 256 / C4;
  const int fh = dst.H / src.H, fw = dst.W / src.W;        // integer fan-out (host-checked)
  const int cs = src.coff + 4 * c4, co = other.coff + 4 * c4, cd = dst.coff + 4 * c4;
  const Xf4 oxf = lhn_load_xf(other, co);
  const int rows = src.N * src.H;
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const int n = row / src.H, h = row - n * src.H;
    const f4 og = other.gate ? *reinterpret_cast<const f4*>(other.gate + (size_t)n * other.cstride + co) : (f4){1.f, 1.f, 1.f, 1.f};
    for (int w = LHN_LANE0(pl, PL); w < src.W; w += PL) {
      f4 g = (f4){0.f, 0.f, 0.f, 0.f};
      for (int a = 0; a < fh; ++a)
        for (int b = 0; b < fw; ++b) {
          const int hd = h * fh + a, wd = w * fw + b;
          const int ho = nearest_src2(hd, other.H, dst.H), wo = nearest_src2(wd, other.W, dst.W);
          const f4 ov = lhn_apply_xf(*reinterpret_cast<const f4*>(other.data + ((size_t)(n * other.H + ho) * other.W + wo) * other.cstride + co), oxf) * og;
          g += *reinterpret_cast<const f4*>(ddst + ((size_t)(n * dst.H + hd) * dst.W + wd) * dst.cstride + cd) * ov;
        }
      float* o = dsrc + ((size_t)row * src.W + w) * src.cstride + cs;
      if (accumulate) g += *reinterpret_cast<const f4*>(o);
      *reinterpret_cast<f4*>(o) = g;
    }
  }
}

// ------------------------------------------------------------------ backward of the bilinear (align_corners) source of a combine
// d(src)[hs, ws] (+)= sum over destination pixels whose taps include (hs, ws) of d(dst) * weight   (gather form)
__device__ __forceinline__ void bil_taps2(int d, int in, int out, int& i0, int& i1, float& w1) {
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
__global__ void __launch_bounds__(256) k_bilinear_bwd(lhn_view src, lhn_view dst, const float* __restrict__ ddst,
                                                      float* __restrict__ dsrc, int accumulate, float out_slope) {
  const int C4 = src.C >> 2, c4 = threadIdx.x % C4, pl = threadIdx.x / C4, PL = 256 / C4;
  const int cs = src.coff + 4 * c4, cd = dst.coff + 4 * c4;
  // destination indices that can touch source index i: d in [ (i-1)*(out-1)/(in-1), (i+1)*(out-1)/(in-1) ] (conservative +-1)
  const float rh = src.H > 1 ? (float)(dst.H - 1) / (float)(src.H - 1) : 0.f, rw = src.W > 1 ? (float)(dst.W - 1) / (float)(src.W - 1) : 0.f;
  const int rows = src.N * src.H;
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const int n = row / src.H, h = row - n * src.H;
    const int hd0 = src.H > 1 ? max(0, (int)floorf((float)(h - 1) * rh) - 1) : 0;
    const int hd1 = src.H > 1 ? min(dst.H - 1, (int)ceilf((float)(h + 1) * rh) + 1) : dst.H - 1;
    for (int w = LHN_LANE0(pl, PL); w < src.W; w += PL) {
      const int wd0 = src.W > 1 ? max(0, (int)floorf((float)(w - 1) * rw) - 1) : 0;
      const int wd1 = src.W > 1 ? min(dst.W - 1, (int)ceilf((float)(w + 1) * rw) + 1) : dst.W - 1;
      f4 g = (f4){0.f, 0.f, 0.f, 0.f};
      for (int hd = hd0; hd <= hd1; ++hd) {
        int a0, a1;
        float ah;
        bil_taps2(hd, src.H, dst.H, a0, a1, ah);
        const float wh = (a0 == h ? 1.f - ah : 0.f) + (a1 == h ? ah : 0.f);
        if (wh == 0.f) continue;
        for (int wd = wd0; wd <= wd1; ++wd) {
          int b0, b1;
          float aw;
          bil_taps2(wd, src.W, dst.W, b0, b1, aw);
          const float ww = (b0 == w ? 1.f - aw : 0.f) + (b1 == w ? aw : 0.f);
          if (ww == 0.f) continue;
          const size_t pd = ((size_t)(n * dst.H + hd) * dst.W + wd) * dst.cstride + cd;
          f4 e = *reinterpret_cast<const f4*>(ddst + pd);
          if (out_slope != 1.f) {
            const f4 o = *reinterpret_cast<const f4*>(dst.data + pd);
            e.x *= o.x > 0.f ? 1.f : out_slope;
            e.y *= o.y > 0.f ? 1.f : out_slope;
            e.z *= o.z > 0.f ? 1.f : out_slope;
            e.w *= o.w > 0.f ? 1.f : out_slope;
          }
          g += e * (wh * ww);
        }
      }
      float* o = dsrc + ((size_t)row * src.W + w) * src.cstride + cs;
      if (accumulate) g += *reinterpret_cast<const f4*>(o);
      *reinterpret_cast<f4*>(o) = g;
    }
  }
}

extern "C" {

int lhn_shuffle2_fwd(const lhn_view* a, const lhn_view* b, const lhn_view* dst, void* stream) {
  LHN_CHECK_ARG(lhn_view_ok(a) && lhn_view_ok(b) && lhn_view_ok(dst) && lhn_no_pend(a) && lhn_no_pend(b), "lhn_shuffle2_fwd: bad views");
  LHN_CHECK_ARG(a->C == b->C && dst->C == 2 * a->C && a->C % 2 == 0 && a->coff % 2 == 0 && b->coff % 2 == 0 && a->cstride % 2 == 0 &&
                    b->cstride % 2 == 0 && dst->C <= 1024,
                "lhn_shuffle2_fwd: channels %d + %d -> %d", a->C, b->C, dst->C);
  LHN_CHECK_ARG(a->N == dst->N && a->H == dst->H && a->W == dst->W && b->N == dst->N && b->H == dst->H && b->W == dst->W,
                "lhn_shuffle2_fwd: geometry");
  hipLaunchKernelGGL(k_shuffle2_fwd, dim3(grid_cap((int64_t)dst->N * dst->H, 8)), dim3(256), 0, (hipStream_t)stream, *a, *b, *dst);
  LHN_CHECK_LAUNCH("lhn_shuffle2_fwd");
  return 0;
}
int lhn_shuffle2_bwd(const lhn_view* a, const lhn_view* b, const lhn_view* dst, const float* ddst, float* da, int acc_a, float* db,
                     int acc_b, void* stream) {
  LHN_CHECK_ARG(lhn_view_ok(a) && lhn_view_ok(b) && lhn_view_ok(dst) && ddst && (da || db), "lhn_shuffle2_bwd: bad args");
  LHN_CHECK_ARG(a->C == b->C && dst->C == 2 * a->C && a->C % 2 == 0 && dst->C <= 1024, "lhn_shuffle2_bwd: channels");
  hipLaunchKernelGGL(k_shuffle2_bwd, dim3(grid_cap((int64_t)dst->N * dst->H, 8)), dim3(256), 0, (hipStream_t)stream, *a, *b, *dst, ddst,
                     da, acc_a, db, acc_b);
  LHN_CHECK_LAUNCH("lhn_shuffle2_bwd");
  return 0;
}
int lhn_ew_mul_bwd(const lhn_view* src, const lhn_view* other, const lhn_view* dst, const float* ddst, float* dsrc, int accumulate,
                   void* stream) {
  LHN_CHECK_ARG(lhn_view_ok(src) && lhn_view_ok(other) && lhn_view_ok(dst) && ddst && dsrc && src->C == dst->C && other->C == dst->C &&
                    src->C <= 1024 && lhn_no_pend(src) && lhn_no_pend(other),
                "lhn_ew_mul_bwd: bad args");
  LHN_CHECK_ARG(dst->H % src->H == 0 && dst->W % src->W == 0, "lhn_ew_mul_bwd: non-integer upsample");
  hipLaunchKernelGGL(k_ew_mul_bwd, dim3(grid_cap((int64_t)src->N * src->H, 8)), dim3(256), 0, (hipStream_t)stream, *src, *other, *dst,
                     ddst, dsrc, accumulate);
  LHN_CHECK_LAUNCH("lhn_ew_mul_bwd");
  return 0;
}
int lhn_bilinear_bwd(const lhn_view* src, const lhn_view* dst, const float* ddst, float* dsrc, int accumulate, float out_slope,
                     void* stream) {
  LHN_CHECK_ARG(lhn_view_ok(src) && lhn_view_ok(dst) && ddst && dsrc && src->C == dst->C && src->C <= 1024 && lhn_no_pend(src),
                "lhn_bilinear_bwd: bad args");
  LHN_CHECK_ARG(src->H <= dst->H && src->W <= dst->W, "lhn_bilinear_bwd: the source must not be larger than the destination");
  hipLaunchKernelGGL(k_bilinear_bwd, dim3(grid_cap((int64_t)src->N * src->H, 8)), dim3(256), 0, (hipStream_t)stream, *src, *dst, ddst,
                     dsrc, accumulate, out_slope);
  LHN_CHECK_LAUNCH("lhn_bilinear_bwd");
  return 0;
}
}  // extern "C"
