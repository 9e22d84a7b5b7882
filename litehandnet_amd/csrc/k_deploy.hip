// Deploy-time re-parameterisation: fold an eval-mode BatchNorm into the kernel/bias of the convolution before it and
// sum parallel branches into one biased convolution (reference repblocks.py:46-73 RepConv, :169-236 RepBlock,
// common.py:68-90 ChannelAttension).  Built without FP contraction: every product/quotient/sum is rounded on its own,
// in the reference's order (sqrt, gamma/std, kernel*t; beta - (mean*gamma)/std); sqrt and divide are the correctly
// rounded expansions, so the fused tensors equal the IEEE-754 float32 evaluation of the reference's formula bit for bit
// (tests pin this against numpy; torch's own CPU sqrt is 1 ulp off on a few inputs).
#include "lhn_common.h"

#pragma clang fp contract(off)

// out_w[co][ci][ky][kx] (=|+=) branch(co,ci,ky,kx) * t[co];  out_b[co] (=|+=) beta - rmean*gamma/std
// branch: w (kb x kb, centred in the k x k window, 0 outside) or, when w == NULL, the identity kernel.
__global__ void __launch_bounds__(256) k_fold_bn(const float* __restrict__ w, int kb, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, const float* __restrict__ rmean,
                                                 const float* __restrict__ rvar, float eps, float* __restrict__ out_w,
                                                 float* __restrict__ out_b, int Cout, int cin_g, int k, int accumulate) {
  const int per = cin_g * k * k;
  const int64_t total = (int64_t)Cout * per;
  const int c0 = k / 2, hb = kb / 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(i / per), r = (int)(i % per);
    const int ci = r / (k * k), ky = (r / k) % k, kx = r % k;
    const float stdv = sqrtf(rvar[co] + eps);
    const float t = gamma[co] / stdv;
    float v;
    bool inside = true;
    if (w) {
      const int by = ky - c0 + hb, bx = kx - c0 + hb;
      inside = by >= 0 && by < kb && bx >= 0 && bx < kb;
      v = inside ? w[((int64_t)(co * cin_g + ci) * kb + by) * kb + bx] : 0.f;
    } else {
      v = (ci == co % cin_g && ky == c0 && kx == c0) ? 1.f : 0.f;
    }
    const float val = inside ? v * t : 0.f;          // zero padding of a small kernel is an exact +0
    out_w[i] = accumulate ? out_w[i] + val : val;
    if (r == 0) {
      const float b = beta[co] - rmean[co] * gamma[co] / stdv;
      out_b[co] = accumulate ? out_b[co] + b : b;
    }
  }
}

extern "C" int lhn_fold_bn(const float* w, int kb, const float* gamma, const float* beta, const float* rmean, const float* rvar,
                           float eps, float* out_w, float* out_b, int Cout, int cin_g, int k, int accumulate, void* stream) {
  LHN_CHECK_ARG(gamma && beta && rmean && rvar && out_w && out_b, "lhn_fold_bn: null pointer");
  LHN_CHECK_ARG(Cout > 0 && cin_g > 0 && k >= 1 && (k & 1) && (w == nullptr || (kb >= 1 && (kb & 1) && kb <= k)),
                "lhn_fold_bn: Cout=%d cin_g=%d k=%d kb=%d", Cout, cin_g, k, kb);
  const int64_t total = (int64_t)Cout * cin_g * k * k;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(k_fold_bn, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, kb, gamma, beta, rmean, rvar, eps, out_w, out_b,
                     Cout, cin_g, k, accumulate);
  LHN_CHECK_LAUNCH("lhn_fold_bn");
  return 0;
}
