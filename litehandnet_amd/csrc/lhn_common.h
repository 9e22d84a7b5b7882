// Shared host/device helpers for liblhn (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <mutex>
#include "../../include/lhn.h"

#define LHN_WAVE 64
// Elementwise kernels map thread -> (channel group c4 = tid % C4, pixel lane pl = tid / C4) with PL = 256 / C4 live lanes.
// C4 = C / 4 need not divide 256 (lite_hrnet.py: C = 20, 40, 80, 160, 320): the left-over threads (pl == PL) start their
// loops at LHN_DEAD, i.e. beyond any extent, and contribute zeros to reductions.
#define LHN_DEAD (1 << 30)
#define LHN_LANE0(pl, PL) ((pl) < (PL) ? (pl) : LHN_DEAD)

void lhn_set_error(const char* fmt, ...);

#define LHN_CHECK_ARG(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      lhn_set_error(__VA_ARGS__);                \
      return 1;                                  \
    }                                            \
  } while (0)

#define LHN_CHECK_LAUNCH(name)                                                    \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) {                                                      \
      lhn_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));       \
      return 2;                                                                   \
    }                                                                             \
  } while (0)

// lhn_view.pend is reserved (round 2's deferred BatchNorm finalize, removed): every entry point wants NULL
static inline int lhn_no_pend(const lhn_view* v) { return !v || !v->pend; }

static inline int lhn_view_ok(const lhn_view* v) {
  return v && v->data && v->N > 0 && v->H > 0 && v->W > 0 && v->C > 0 && v->coff >= 0 &&
         v->coff + v->C <= v->cstride && (v->cstride % 4) == 0 && (v->coff % 4) == 0 && (v->C % 4) == 0;
}

// Host-side state is PER DEVICE (the library is re-entrant across devices: one process per GPU, or nn.DataParallel
// threads each bound to their own device, test.py:81): the CU count and every kernel's one-time attributes
// (hipFuncSetAttribute is a per-device setting) live in tables indexed by the calling thread's current device and are
// initialised under std::call_once.
#define LHN_MAX_DEVICES 16
static inline int lhn_device_slot() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0) d = 0;
  return d % LHN_MAX_DEVICES;
}
int lhn_num_cus();   // lhn_api.cpp: CUs of the current device; grid caps for persistent grid-stride kernels
bool lhn_deterministic_mode();   // LHN_DETERMINISTIC=1, see lhn_api.cpp

#ifdef __HIPCC__
// Workgroups of `kernel` (256 threads, `dyn_lds` bytes of dynamic LDS) that are resident on one CU at the same time: the
// occupancy API, never more than the LDS arithmetic with the kernel's static segment included (a persistent grid that
// over-estimates this by one runs a second, half-empty round -- measured: 3 x 54,784 B does not fit 160 KB).
template <typename F>
static inline int lhn_resident_per_cu(F kernel, size_t dyn_lds, int cap) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, 256, dyn_lds) != hipSuccess || n < 1) n = 1;
  hipFuncAttributes a;
  if (hipFuncGetAttributes(&a, reinterpret_cast<const void*>(kernel)) == hipSuccess) {
    size_t per = (dyn_lds + a.sharedSizeBytes + 1023) / 1024 * 1024;
    const int by_lds = per ? (int)((160 * 1024) / per) : n;
    if (by_lds < n) n = by_lds;
  }
  if (n > cap) n = cap;
  return n < 1 ? 1 : n;
}

// One-time, per-device setup of a kernel: dynamic-LDS opt-in + resident workgroups per CU.  `static LhnKernelCfg cfg;`
// at the launch site (one per template instance); returns false when the LDS reservation is refused.
struct LhnKernelCfg {
  std::once_flag once[LHN_MAX_DEVICES];
  int per_cu[LHN_MAX_DEVICES];
  bool ok[LHN_MAX_DEVICES];
};
template <typename F>
static inline bool lhn_kernel_cfg(LhnKernelCfg& c, F kernel, size_t dyn_lds, int cap, int* per_cu) {
  const int d = lhn_device_slot();
  std::call_once(c.once[d], [&] {
    c.ok[d] = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)dyn_lds) == hipSuccess;
    c.per_cu[d] = lhn_resident_per_cu(kernel, dyn_lds, cap);
  });
  if (per_cu) *per_cu = c.per_cu[d];
  return c.ok[d];
}

// ---------------------------------------------------------------- device side
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float lhn_lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }
// SiLU (models/pose_hg_ms_att.py:82,87): only the elementwise combine applies it (out_slope == LHN_SLOPE_SILU)
__device__ __forceinline__ float lhn_silu(float v) { return v / (1.f + expf(-v)); }
__device__ __forceinline__ float lhn_silu_grad(float v) {
  const float s = 1.f / (1.f + expf(-v));
  return s * (1.f + v * (1.f - s));
}

// pending transform of 4 consecutive channels of a view (absolute channel c, multiple of 4)
struct Xf4 {
  f4 sc, sh, sl;
};
__device__ __forceinline__ Xf4 lhn_load_xf(const lhn_view& v, int c_abs) {
  Xf4 t;
  if (v.table) {
    t.sc = *reinterpret_cast<const f4*>(v.table + c_abs);
    t.sh = *reinterpret_cast<const f4*>(v.table + v.cstride + c_abs);
    t.sl = *reinterpret_cast<const f4*>(v.table + 2 * v.cstride + c_abs);
  } else {
    t.sc = (f4){1.f, 1.f, 1.f, 1.f};
    t.sh = (f4){0.f, 0.f, 0.f, 0.f};
    t.sl = (f4){1.f, 1.f, 1.f, 1.f};
  }
  return t;
}
__device__ __forceinline__ f4 lhn_apply_xf(f4 raw, const Xf4& t) {
  f4 u = raw * t.sc + t.sh;
  f4 r;
  r.x = lhn_lrelu(u.x, t.sl.x);
  r.y = lhn_lrelu(u.y, t.sl.y);
  r.z = lhn_lrelu(u.z, t.sl.z);
  r.w = lhn_lrelu(u.w, t.sl.w);
  return r;
}
// consumed value of 4 channels at pixel index pix (= (n*H+h)*W+w), image n
__device__ __forceinline__ f4 lhn_load_val(const lhn_view& v, const Xf4& t, int64_t pix, int n, int c_abs) {
  f4 raw = *reinterpret_cast<const f4*>(v.data + pix * v.cstride + c_abs);
  f4 r = lhn_apply_xf(raw, t);
  if (v.gate) r *= *reinterpret_cast<const f4*>(v.gate + (int64_t)n * v.cstride + c_abs);
  return r;
}

// d(loss)/d(BN output u) and d(loss)/d(raw y) for 4 channels of a conv output, given the gradient
// w.r.t. the consumed value.  eff_dz = gate*dz + sum_bins dpool ;  du = eff_dz * lrelu'(u) ;
// dy = A*du + B*y + C  (coef == NULL: dy = du)
struct Gr4 {
  f4 A, B, Cc;
};
__device__ __forceinline__ Gr4 lhn_load_coef(const lhn_gradview& g, int cstride, int c_abs) {
  Gr4 r;
  if (g.coef) {
    r.A = *reinterpret_cast<const f4*>(g.coef + c_abs);
    r.B = *reinterpret_cast<const f4*>(g.coef + cstride + c_abs);
    r.Cc = *reinterpret_cast<const f4*>(g.coef + 2 * cstride + c_abs);
  } else {
    r.A = (f4){1.f, 1.f, 1.f, 1.f};
    r.B = (f4){0.f, 0.f, 0.f, 0.f};
    r.Cc = (f4){0.f, 0.f, 0.f, 0.f};
  }
  return r;
}
// adaptive-avg-pool(3x3) bin membership: bin i covers [floor(i*S/3), ceil((i+1)*S/3))
__device__ __forceinline__ int lhn_bin_lo(int i, int S) { return (i * S) / 3; }
__device__ __forceinline__ int lhn_bin_hi(int i, int S) { return ((i + 1) * S + 2) / 3; }

// The pooled gradient is stored per SEGMENT, not per bin: the three (possibly overlapping) bins cut each axis into at
// most five segments -- {bin0}, {bin0,bin1}, {bin1}, {bin1,bin2}, {bin2} -- and the attention backward writes
// dpool[n][sh*5+sw][c] = sum of (d loss / d pooled[bin]) / |bin| over the bins of segment (sh, sw).  A consumer then needs
// ONE branch-free load per pixel instead of a divergent loop over up to four bins (H, W >= 2).
#define LHN_DPOOL_SLOTS 25
__device__ __forceinline__ int lhn_pool_seg(int h, int S) {
  return (h >= lhn_bin_lo(1, S)) + (h >= lhn_bin_hi(0, S)) + (h >= lhn_bin_lo(2, S)) + (h >= lhn_bin_hi(1, S));
}
__device__ __forceinline__ f4 lhn_dpool_sum(const lhn_gradview& g, const lhn_view& v, int n, int h, int w, int c_abs) {
  const int slot = lhn_pool_seg(h, v.H) * 5 + lhn_pool_seg(w, v.W);
  return *reinterpret_cast<const f4*>(g.dpool + ((int64_t)n * LHN_DPOOL_SLOTS + slot) * v.cstride + c_abs);
}
// writer side: d[t] = d loss / d pooled[bin t] / |bin t| for the 9 bins of one (n, channel) -> the 25 segment sums
__device__ __forceinline__ void lhn_dpool_store(float* dpool, int64_t n, int cs, int c_abs, const float (&d)[9]) {
#pragma unroll
  for (int sh = 0; sh < 5; ++sh)
#pragma unroll
    for (int sw = 0; sw < 5; ++sw) {
      float v = 0.f;
#pragma unroll
      for (int bi = 0; bi < 3; ++bi)
#pragma unroll
        for (int bj = 0; bj < 3; ++bj) {
          const bool rin = (sh == 2 * bi) || (sh == 2 * bi - 1) || (sh == 2 * bi + 1);
          const bool cin = (sw == 2 * bj) || (sw == 2 * bj - 1) || (sw == 2 * bj + 1);
          if (rin && cin) v += d[bi * 3 + bj];
        }
      dpool[(n * LHN_DPOOL_SLOTS + sh * 5 + sw) * cs + c_abs] = v;
    }
}
// returns du (gradient at the BN output) and, through *val_out, nothing else; raw = y
__device__ __forceinline__ f4 lhn_grad_du(const lhn_view& v, const lhn_gradview& g, const Xf4& t, f4 raw, f4 dz,
                                          int n, int h, int w, int c_abs) {
  f4 e = dz;
  if (v.gate) e *= *reinterpret_cast<const f4*>(v.gate + (int64_t)n * v.cstride + c_abs);
  f4 u = raw * t.sc + t.sh;
  f4 du;
  du.x = e.x * (u.x > 0.f ? 1.f : t.sl.x);
  du.y = e.y * (u.y > 0.f ? 1.f : t.sl.y);
  du.z = e.z * (u.z > 0.f ? 1.f : t.sl.z);
  du.w = e.w * (u.w > 0.f ? 1.f : t.sl.w);
  if (g.dpool) {
    f4 dp = lhn_dpool_sum(g, v, n, h, w, c_abs);   // pooled value is the *post-activation* value
    du.x += dp.x * (u.x > 0.f ? 1.f : t.sl.x);
    du.y += dp.y * (u.y > 0.f ? 1.f : t.sl.y);
    du.z += dp.z * (u.z > 0.f ? 1.f : t.sl.z);
    du.w += dp.w * (u.w > 0.f ? 1.f : t.sl.w);
  }
  return du;
}

// Lean dy for the MFMA loaders: gate (1 if none) is passed in, the channel-attention pooled gradient is
// handled by the caller on a separate (rare) path.
__device__ __forceinline__ f4 lhn_dy_fast(const Xf4& t, const Gr4& gr, f4 raw, f4 dz, f4 gate) {
  const f4 u = raw * t.sc + t.sh;
  const f4 dl = (f4){u.x > 0.f ? 1.f : t.sl.x, u.y > 0.f ? 1.f : t.sl.y, u.z > 0.f ? 1.f : t.sl.z, u.w > 0.f ? 1.f : t.sl.w};
  return gr.A * (dz * gate * dl) + gr.B * raw + gr.Cc;
}

// ---- last-block-arrives hand-off for ATOMIC accumulators.  The only bytes handed over are sums built with
// device-scope atomic adds, which execute at the memory side and never sit in an L1/L2 (MI355X_MICROARCH
// "Global float atomics"), so no cache write-back / invalidate is needed (a release fence here costs a
// buffer_wbl2 per workgroup and made every conv slower than the separate finalize launch it replaced):
// each wave waits until its own atomics are acknowledged (vmcnt(0)), the block meets at a barrier, ONE lane
// takes a ticket with a device-scope atomic; the block that draws the last ticket reads the sums with
// device-scope (sc1, L1-bypassing) loads.  Nothing else in the launch ever reads those addresses.
// Tickets in TWO levels (counter[0] = top, counter[1 + g] = group g = block % 32): returning atomics on ONE word are served at
// ~88 per us, so 1,024 workgroups finishing together queued 12 us on a single word (round 1: fused finalize slower than the
// launch it replaced); 32 group words take their tickets in parallel, the last of each group takes one of 32 top tickets.
#define LHN_TICKET_WORDS 33
__device__ __forceinline__ bool lhn_last_block(unsigned* counter) {
  __shared__ int s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned nb = gridDim.x * gridDim.y, b = blockIdx.x + blockIdx.y * gridDim.x;
    const unsigned g = b & 31u, ng = nb < 32u ? nb : 32u, gsize = (nb - g + 31u) >> 5;
    int last = 0;
    if (__hip_atomic_fetch_add(counter + 1 + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1u)
      last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ng - 1u;
    s_last = last;
  }
  __syncthreads();
  return s_last != 0;
}
__device__ __forceinline__ double lhn_ld_agent(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// train-mode BatchNorm finalize by one whole block (same arithmetic as k_bn_finalize)
__device__ __forceinline__ void lhn_bn_finalize_block(const lhn_bnfin& f, const double* stats) {
  const int C = f.C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    double s1 = 0, s2 = 0;
    for (int r = 0; r < LHN_STAT_REPLICAS; ++r) {
      s1 += lhn_ld_agent(stats + (size_t)r * 2 * C + c);
      s2 += lhn_ld_agent(stats + (size_t)r * 2 * C + C + c);
    }
    const double mean = s1 / f.count;
    double var = s2 / f.count - mean * mean;
    if (var < 0) var = 0;
    if (f.running_mean) {
      const double bm = mean + (f.conv_bias ? (double)f.conv_bias[c] : 0.0);
      f.running_mean[c] = (float)((1.0 - (double)f.momentum) * (double)f.running_mean[c] + (double)f.momentum * bm);
      const double unb = f.count > 1 ? var * f.count / (f.count - 1.0) : var;
      f.running_var[c] = (float)((1.0 - (double)f.momentum) * (double)f.running_var[c] + (double)f.momentum * unb);
    }
    const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
    const float g = f.gamma ? f.gamma[c] : 1.f, b = f.beta ? f.beta[c] : 0.f;
    const float sc = g * invstd;
    f.table[f.coff + c] = sc;
    f.table[f.cstride + f.coff + c] = b - (float)mean * sc;
    f.table[2 * f.cstride + f.coff + c] = f.slope;
    if (f.save_mean_invstd) {
      f.save_mean_invstd[c] = (float)mean;
      f.save_mean_invstd[C + c] = invstd;
    }
  }
  if (f.num_batches_tracked && threadIdx.x == 0) f.num_batches_tracked[0] += 1;
}
// the same by a whole block with the replica fold spread over its threads (blockDim.x / C groups of replicas per channel meet in
// `part`, LDS for 2 * blockDim.x doubles that nobody else is using any more): one memory latency instead of 64 dependent loads
__device__ __forceinline__ void lhn_bn_finalize_block_par(const lhn_bnfin& f, const double* stats, double* part) {
  const int C = f.C, nt = blockDim.x, tid = threadIdx.x;
  int G = nt / C;
  G = G < 1 ? 1 : (G > LHN_STAT_REPLICAS ? LHN_STAT_REPLICAS : G);
  __syncthreads();      // (part may alias LDS the block was still reading)
  for (int c0 = 0; c0 < C; c0 += nt) {
    const int g = G > 1 ? tid / C : 0, c = c0 + (G > 1 ? tid - g * C : tid);
    if (c0) __syncthreads();
    if (g < G && c < C) {
      double s1 = 0, s2 = 0;
#pragma unroll 4
      for (int r = g; r < LHN_STAT_REPLICAS; r += G) {
        s1 += lhn_ld_agent(stats + (size_t)r * 2 * C + c);
        s2 += lhn_ld_agent(stats + (size_t)r * 2 * C + C + c);
      }
      part[2 * tid] = s1;
      part[2 * tid + 1] = s2;
    }
    __syncthreads();
    if (g != 0 || c >= C) continue;
    double s1 = 0, s2 = 0;
    for (int k = 0; k < G; ++k) {
      s1 += part[2 * (k * C + tid)];
      s2 += part[2 * (k * C + tid) + 1];
    }
    const double mean = s1 / f.count;
    double var = s2 / f.count - mean * mean;
    if (var < 0) var = 0;
    if (f.running_mean) {
      const double bm = mean + (f.conv_bias ? (double)f.conv_bias[c] : 0.0);
      f.running_mean[c] = (float)((1.0 - (double)f.momentum) * (double)f.running_mean[c] + (double)f.momentum * bm);
      const double unb = f.count > 1 ? var * f.count / (f.count - 1.0) : var;
      f.running_var[c] = (float)((1.0 - (double)f.momentum) * (double)f.running_var[c] + (double)f.momentum * unb);
    }
    const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
    const float gm = f.gamma ? f.gamma[c] : 1.f, b = f.beta ? f.beta[c] : 0.f;
    const float sc = gm * invstd;
    f.table[f.coff + c] = sc;
    f.table[f.cstride + f.coff + c] = b - (float)mean * sc;
    f.table[2 * f.cstride + f.coff + c] = f.slope;
    if (f.save_mean_invstd) {
      f.save_mean_invstd[c] = (float)mean;
      f.save_mean_invstd[C + c] = invstd;
    }
  }
  if (f.num_batches_tracked && tid == 0) f.num_batches_tracked[0] += 1;
}
__device__ __forceinline__ void lhn_bn_bwd_finalize_block(const lhn_bnbwdfin& f, const double* sums, const float* save) {
  const int C = f.C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    double db = 0, dg = 0;
    for (int r = 0; r < LHN_STAT_REPLICAS; ++r) {
      db += lhn_ld_agent(sums + (size_t)r * 2 * C + c);
      dg += lhn_ld_agent(sums + (size_t)r * 2 * C + C + c);
    }
    const double mean = save[c], inv = save[C + c], s = (double)(f.gamma ? f.gamma[c] : 1.f) * inv;
    f.coef[f.coff + c] = (float)s;
    f.coef[f.cstride + f.coff + c] = (float)(-s * inv * dg / f.count);
    f.coef[2 * f.cstride + f.coff + c] = (float)(-s * db / f.count + s * inv * mean * dg / f.count);
    if (f.dgamma) f.dgamma[c] += (float)dg;
    if (f.dbeta) f.dbeta[c] += (float)db;
  }
}
static __device__ __forceinline__ lhn_bnfin lhn_nofin() {
  lhn_bnfin f;
  f.counter = nullptr;
  return f;
}

// Block-wide per-channel sums for the thread layout (c4 = tid % C4 float4 channel groups, pixel lane = tid / C4) with
// C4 in {8, 16, 32}: lanes of a wave that share c4 meet by xor-shuffles, the four waves through `red` (>= 8*C4 float4 of
// LDS that nobody else is using), then 8*C4 threads add one double each into st0[4*c4+j] (sums) / st1[4*c4+j] (second sums).
// Replaces a C4-thread serial loop over 256/C4 LDS rows that cost 4-10 us per launch.
__device__ __forceinline__ void lhn_block_stat_atomics(f4 s, f4 q, int C4, f4* red, double* st0, double* st1, int nvalid = 64) {
  for (int o = C4; o < 64; o <<= 1) {
    s.x += __shfl_xor(s.x, o, 64); s.y += __shfl_xor(s.y, o, 64); s.z += __shfl_xor(s.z, o, 64); s.w += __shfl_xor(s.w, o, 64);
    q.x += __shfl_xor(q.x, o, 64); q.y += __shfl_xor(q.y, o, 64); q.z += __shfl_xor(q.z, o, 64); q.w += __shfl_xor(q.w, o, 64);
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __syncthreads();
  if (lane < C4) {
    red[(wave * C4 + lane) * 2] = s;
    red[(wave * C4 + lane) * 2 + 1] = q;
  }
  __syncthreads();
  if (tid < 8 * C4) {
    const int kind = tid / (4 * C4), r = tid - kind * 4 * C4, cc = r >> 2, jj = r & 3;
    double v = 0;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) {
      const f4 a = red[(wv * C4 + cc) * 2 + kind];
      v += (double)(jj == 0 ? a.x : jj == 1 ? a.y : jj == 2 ? a.z : a.w);
    }
    if (cc < nvalid) atomicAdd((kind ? st1 : st0) + 4 * cc + jj, v);      // nvalid: channel groups that exist (tail of C % 32)
  }
}

// lhn_load_xf from an explicit table pointer; tab == NULL = identity
__device__ __forceinline__ Xf4 lhn_load_xf_t(const float* tab, int cstride, int c_abs) {
  Xf4 t;
  if (tab) {
    t.sc = *reinterpret_cast<const f4*>(tab + c_abs);
    t.sh = *reinterpret_cast<const f4*>(tab + cstride + c_abs);
    t.sl = *reinterpret_cast<const f4*>(tab + 2 * cstride + c_abs);
  } else {
    t.sc = (f4){1.f, 1.f, 1.f, 1.f};
    t.sh = (f4){0.f, 0.f, 0.f, 0.f};
    t.sl = (f4){1.f, 1.f, 1.f, 1.f};
  }
  return t;
}

// Per-tile BatchNorm partial sums, SHIFTED: within one tile a thread accumulates sum(v - K) and sum((v - K)^2) in fp32 with
// K = the first value it saw in this tile (differences of the order of sigma: no cancellation, full fp32 relative accuracy
// even when |mean| = 1000 sigma), and un-shifts in double when the tile is done:
//   sum v   = s + n K            sum v^2 = q + 2 K s + n K^2
struct TileStat {
  float k, s, q;
  int n;
  __device__ __forceinline__ void reset() { n = 0; k = s = q = 0.f; }
  __device__ __forceinline__ void add(float v) {
    if (n == 0) k = v;
    const float d = v - k;
    s += d;
    q += d * d;
    ++n;
  }
  __device__ __forceinline__ void flush(double& sum, double& sq) const {
    const double K = (double)k, S = (double)s;
    sum += S + (double)n * K;
    sq += (double)q + 2.0 * K * S + (double)n * K * K;
  }
};

// float4 form: sd / qd (double[4]) += the un-shifted sums of n values accumulated as s = sum(v - k), q = sum((v - k)^2)
__device__ __forceinline__ void lhn_unshift4(double (&sd)[4], double (&qd)[4], f4 s, f4 q, f4 k, int n) {
  const float sv[4] = {s.x, s.y, s.z, s.w}, qv[4] = {q.x, q.y, q.z, q.w}, kv[4] = {k.x, k.y, k.z, k.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const double K = (double)kv[j], S = (double)sv[j];
    sd[j] += S + (double)n * K;
    qd[j] += (double)qv[j] + 2.0 * K * S + (double)n * K * K;
  }
}

// The same reduction for DOUBLE per-thread sums (convolution epilogues promote their per-tile fp32 partials to double: a
// running fp32 sum of squares over thousands of pixels loses the variance when |mean| >> sigma).  red: >= 32*C4 doubles.
__device__ __forceinline__ void lhn_block_stat_atomics_d(const double (&s)[4], const double (&q)[4], int C4, double* red,
                                                         double* st0, double* st1, int nvalid = 64) {
  double v[8] = {s[0], s[1], s[2], s[3], q[0], q[1], q[2], q[3]};
  for (int o = C4; o < 64; o <<= 1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += __shfl_xor(v[i], o, 64);
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __syncthreads();
  if (lane < C4) {
#pragma unroll
    for (int i = 0; i < 8; ++i) red[(wave * C4 + lane) * 8 + i] = v[i];
  }
  __syncthreads();
  if (tid < 8 * C4) {
    const int kind = tid / (4 * C4), r = tid - kind * 4 * C4, cc = r >> 2, jj = r & 3;
    double t = 0;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) t += red[(wv * C4 + cc) * 8 + kind * 4 + jj];
    if (cc < nvalid) atomicAdd((kind ? st1 : st0) + 4 * cc + jj, t);
  }
}

// ---- reader-side BatchNorm-backward sums (lhn_bnsum), elementwise thread layout (c4 = tid % C4 float4 channel groups of the
// reader's input view, pixel lane = tid / C4; dead lanes carry zeros): block reduction + one double atomic per channel and
// block into the producer's replica (blockIdx.x % LHN_STAT_REPLICAS).  red: >= 512 float4 of LDS nobody else is using.
__device__ __forceinline__ void lhn_bns_flush(const lhn_bnsum& b, f4 s, f4 q, int C4, f4* red) {
  double* st = b.sums + (size_t)(blockIdx.x % LHN_STAT_REPLICAS) * 2 * b.C + b.coff;
  if (C4 <= 32 && (C4 & (C4 - 1)) == 0) {
    lhn_block_stat_atomics(s, q, C4, red, st, st + b.C);
  } else {
    const int PL = 256 / C4;
    __syncthreads();
    red[threadIdx.x * 2] = s;
    red[threadIdx.x * 2 + 1] = q;
    __syncthreads();
    if ((int)threadIdx.x < C4) {
      double sd[4] = {0, 0, 0, 0}, qd[4] = {0, 0, 0, 0};
      for (int j = 0; j < PL; ++j) {
        const f4 a = red[(j * C4 + threadIdx.x) * 2], c = red[(j * C4 + threadIdx.x) * 2 + 1];
        sd[0] += a.x; sd[1] += a.y; sd[2] += a.z; sd[3] += a.w;
        qd[0] += c.x; qd[1] += c.y; qd[2] += c.z; qd[3] += c.w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        atomicAdd(st + 4 * threadIdx.x + j, sd[j]);
        atomicAdd(st + b.C + 4 * threadIdx.x + j, qd[j]);
      }
    }
  }
}
// derivative of the pending activation at the raw value
__device__ __forceinline__ f4 lhn_dact_xf(f4 raw, const Xf4& t) {
  const f4 u = raw * t.sc + t.sh;
  return (f4){u.x > 0.f ? 1.f : t.sl.x, u.y > 0.f ? 1.f : t.sl.y, u.z > 0.f ? 1.f : t.sl.z, u.w > 0.f ? 1.f : t.sl.w};
}

__device__ __forceinline__ float lhn_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double lhn_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
#endif
