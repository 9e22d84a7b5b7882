// Dense 3x3 convolution (BasicBlock / BottleNeck of variant A).  Placeholder until the MFMA implicit-GEMM
// kernels land: the entry points exist so the ABI is complete and fail loudly.
#include "lhn_common.h"

extern "C" int lhn_conv_kxk_fwd(const lhn_view* x, const float* w, const lhn_view* y, double* stats, int stride,
                                void* stream) {
  (void)x; (void)w; (void)y; (void)stats; (void)stride; (void)stream;
  lhn_set_error("lhn_conv_kxk_fwd: dense 3x3 convolution is not built yet");
  return 3;
}
extern "C" int lhn_conv_kxk_bwd(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, float* dx,
                                int dx_accumulate, float* dw, int stride, int nrep, int64_t rep_stride, void* stream) {
  (void)nrep; (void)rep_stride;
  (void)x; (void)w; (void)y; (void)gy; (void)dx; (void)dx_accumulate; (void)dw; (void)stride; (void)stream;
  lhn_set_error("lhn_conv_kxk_bwd: dense 3x3 convolution is not built yet");
  return 3;
}
