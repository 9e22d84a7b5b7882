// Static plan executor: one call enqueues a whole forward or backward pass on a HIP stream.
// The plan (buffers + op lists) is produced by the Python mirror of the reference's module tree;
// this file only resolves offsets into the workspace arena / parameter array and calls the launchers.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "lhn_common.h"

// reader-side BatchNorm sums (lhn_bnsum) of an op: byte offsets of the producer's sums / saved statistics, channels, offset
static lhn_bnsum mkbns(void* ws, int64_t sums_off, int64_t save_off, int C, int coff) {
  lhn_bnsum b;
  memset(&b, 0, sizeof(b));
  if (sums_off >= 0 && save_off >= 0 && C > 0) {
    b.sums = reinterpret_cast<double*>(static_cast<char*>(ws) + sums_off);
    b.save = reinterpret_cast<const float*>(static_cast<char*>(ws) + save_off);
    b.C = C;
    b.coff = coff;
  }
  return b;
}

enum {
  OP_STEM = 1, OP_PW = 2, OP_DW = 3, OP_KXK = 4, OP_FINALIZE = 5, OP_EW = 6, OP_MAXPOOL = 7, OP_AVGPOOL = 8,
  OP_CA_MLP = 9, OP_TABLE_FILL = 10, OP_MEMSET = 11, OP_ATT_MLP = 12, OP_SE_MLP = 13, OP_SHUFFLE = 14,
  OP_STEM_BWD = 101, OP_PW_BWD = 102, OP_DW_BWD = 103, OP_KXK_BWD = 104, OP_BN_BWD = 105, OP_EW_BWD = 106,
  OP_MAXPOOL_BWD = 107, OP_AVGPOOL_BWD = 108, OP_GATE_REDUCE = 109, OP_CA_MLP_BWD = 110, OP_ATT_MLP_BWD = 111, OP_SE_MLP_BWD = 112,
  OP_SHUFFLE_BWD = 113,
};

// One captured launch sequence (hipGraph) of a phase for one set of pointers.
struct GraphEntry {
  int phase, training, nrep, seen;
  int64_t rstr;
  void *ws, *io0, *io1;
  uint64_t phash, ghash;
  hipGraphExec_t exec;
  uint64_t last_use;
};

struct Plan {
  std::vector<lhn_buf> bufs;
  std::vector<lhn_op> fwd, bwd;
  std::vector<GraphEntry> graphs;      // LHN_GRAPH=1: replayed instead of ~150-400 individual launches
  uint64_t tick = 0;
  int graph_misses = 0;
};

static inline char* at(void* ws, int64_t off) { return off < 0 ? nullptr : static_cast<char*>(ws) + off; }

static lhn_view mkview(const Plan* P, void* ws, int buf, int coff, int C, bool with_gate = true) {
  const lhn_buf& b = P->bufs[buf];
  lhn_view v;
  v.pend = nullptr;      // (reserved)
  v.data = reinterpret_cast<float*>(at(ws, b.data_off));
  v.table = reinterpret_cast<const float*>(at(ws, b.table_off));
  v.gate = with_gate ? reinterpret_cast<const float*>(at(ws, b.gate_off)) : nullptr;
  v.N = b.N; v.H = b.H; v.W = b.W;
  v.cstride = b.C; v.coff = coff; v.C = C;
  return v;
}
static lhn_gradview mkgrad(const Plan* P, void* ws, int buf, bool use_coef) {
  const lhn_buf& b = P->bufs[buf];
  lhn_gradview g;
  g.dz = reinterpret_cast<const float*>(at(ws, b.grad_off));
  g.dpool = reinterpret_cast<const float*>(at(ws, b.dpool_off));
  g.coef = use_coef ? reinterpret_cast<const float*>(at(ws, b.coef_off)) : nullptr;
  return g;
}
template <typename T>
static inline T* prm(void* const* arr, int idx) { return idx < 0 ? nullptr : static_cast<T*>(arr[idx]); }
// BatchNorm slices of a gated buffer (lhn_bn_slices): packed (first channel << 16 | channels) in pk[0..1] (0 = none), byte offsets
// of the saved statistics in sv[0..1] and of the backward sums in sm[0..1] (NULL: sv only)
static lhn_bn_slices mkslices(void* ws, const int32_t* pk, const int64_t* sv, const int64_t* sm) {
  lhn_bn_slices s;
  memset(&s, 0, sizeof(s));
  for (int k = 0; k < 2; ++k)
    if (pk[k] > 0 && sv[k] >= 0) {
      const int j = s.n++;
      s.lo[j] = pk[k] >> 16;
      s.C[j] = pk[k] & 0xffff;
      s.save[j] = reinterpret_cast<const float*>(at(ws, sv[k]));
      s.sums[j] = (sm && sm[k] >= 0) ? reinterpret_cast<double*>(at(ws, sm[k])) : nullptr;
    }
  return s;
}

// conv op with a trailing BatchNorm: p[2..6] = gamma, beta, running_mean, running_var, num_batches_tracked,
// ws[1] = save(mean,invstd), ws[2] = arrival counter, f[0..2] = eps, momentum, slope
static bool conv_has_bn(const lhn_op& o) { return o.p[2] >= 0 || o.p[3] >= 0 || o.ws[1] >= 0; }
static lhn_bnfin mkfin(const Plan* P, void* ws, const lhn_op& o, void* const* params) {
  const lhn_buf& b = P->bufs[o.out_buf];
  lhn_bnfin f;
  f.counter = reinterpret_cast<uint32_t*>(at(ws, o.ws[2]));
  f.gamma = prm<const float>(params, o.p[2]);
  f.beta = prm<const float>(params, o.p[3]);
  f.running_mean = prm<float>(params, o.p[4]);
  f.running_var = prm<float>(params, o.p[5]);
  f.num_batches_tracked = prm<int64_t>(params, o.p[6]);
  f.table = reinterpret_cast<float*>(at(ws, b.table_off));
  f.save_mean_invstd = reinterpret_cast<float*>(at(ws, o.ws[1]));
  f.count = (double)b.N * b.H * b.W;
  f.cstride = b.C; f.coff = o.out_coff; f.C = o.out_C;
  f.eps = o.f[0]; f.momentum = o.f[1]; f.slope = o.f[2];
  f.conv_bias = prm<const float>(params, o.p[1]);   // biased conv + BN: the bias lives in the finalize only
  return f;
}
// LHN_FUSE_FINALIZE=1 (default 0): the last workgroup of a convolution folds the statistics and writes the table (lhn_bnfin) instead
// of a separate 1-workgroup launch.  Round 1 measured it SLOWER (one returning ticket atomic per workgroup on one word, ~88 tickets
// per us: 13.5 vs 12.9 ms per step of variant B); round 3 takes the tickets in two levels (32 group words + 1 top word) and folds
// the replicas with the whole block (lhn_last_block / lhn_bn_finalize_block_par).  Same box, alternating: variant B forward 2.550 ->
// 2.542 ms (97 instead of 149 launches), step 8.05 -> 8.06; Lite-HRNet step 35.85 -> 36.61; A 19.30 -> 19.42.  A finalize is its
// latency chain either way, so the separate launch stays the default.  Whole-plan runs only (SyncBatchNorm splits the op).
static bool fuse_finalize() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("LHN_FUSE_FINALIZE");
    v = (e && e[0] == '1') ? 1 : 0;
  }
  return v == 1;
}
// creal: channels the BatchNorm really has when the convolution's output view is padded to a multiple of 4 (f.C then is the
// layout of the statistics); repeat: the reference evaluates some units twice per forward (lite_hrnet.py:192-197), which
// moves their running statistics twice -- the table is the same both times.
static int sep_finalize(const lhn_bnfin& f, const double* stats, int training, void* stream, int creal = 0, int repeat = 1) {
  int rc = 0;
  for (int r = 0; r < (training ? (repeat < 1 ? 1 : repeat) : 1) && !rc; ++r)
    rc = lhn_bn_finalize2(training ? stats : nullptr, f.gamma, f.beta, f.running_mean, f.running_var,
                          training ? f.num_batches_tracked : nullptr, f.table, f.cstride, f.coff, creal > 0 ? creal : f.C, f.C,
                          training ? f.save_mean_invstd : nullptr, f.count, f.eps, f.momentum, f.slope, training, f.conv_bias, stream);
  return rc;
}

extern "C" {

void* lhn_plan_create(const lhn_buf* bufs, int nbufs, const lhn_op* fwd, int nfwd, const lhn_op* bwd, int nbwd) {
  if (!bufs || nbufs <= 0 || !fwd || nfwd <= 0) {
    lhn_set_error("lhn_plan_create: empty plan");
    return nullptr;
  }
  Plan* p = new Plan();
  p->bufs.assign(bufs, bufs + nbufs);
  p->fwd.assign(fwd, fwd + nfwd);
  if (bwd && nbwd > 0) p->bwd.assign(bwd, bwd + nbwd);
  for (const auto* lst : {&p->fwd, &p->bwd})
    for (const lhn_op& o : *lst) {
      bool ok = o.out_buf < nbufs;
      for (int k = 0; k < 3; ++k) ok = ok && o.in_buf[k] < nbufs;
      if (!ok) {
        lhn_set_error("lhn_plan_create: op kind %d references buffer out of range", o.kind);
        delete p;
        return nullptr;
      }
    }
  return p;
}

void lhn_plan_destroy(void* plan) {
  Plan* p = static_cast<Plan*>(plan);
  if (!p) return;
  for (GraphEntry& g : p->graphs)
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
  delete p;
}

// Ops are numbered in HALF-steps: step 2*i = the op's main launches, step 2*i+1 = its statistics consumer (BatchNorm
// finalize, second half of an attention op).  SyncBatchNorm runs [.., 2*i] / all-reduce / [2*i+1, ..]; a plain run is
// the whole range.  count_scale = world size (statistics are over N*world samples), pgrad_scale = 1/world for the
// d(gamma), d(beta) that come out of globally reduced sums.
static int run_ops(const Plan* P, int phase, void* ws, void* const* params, void* const* grads, void* const* io, int mode,
                   int nrep, int64_t rstr, void* stream, size_t sb = 0, size_t se = (size_t)-1, double cscale = 1.0,
                   float pscale = 1.f) {
  // mode bit 0 = training (batch statistics); bit 1 (eval only, LHN_RUN_TABLES_CURRENT) = the per-buffer (scale, shift,
  // slope) tables already hold the running-statistics BatchNorms / deployed biases of the current parameters: skip the
  // launches that only rebuild them (52 of variant B's ~200 forward launches)
  const int training = mode & 1;
  const bool skip_tables = (mode & 2) != 0 && !training;
  const std::vector<lhn_op>& ops = phase == 0 ? P->fwd : P->bwd;
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = 0;
  const bool whole = (sb == 0 && se >= 2 * ops.size());
  for (size_t oi = 0; oi < ops.size() && rc == 0; ++oi) {
    const lhn_op& o = ops[oi];
    const bool deferred = false;      // (round 2's deferred finalize is gone)
    // fused finalize (last workgroup of the convolution): whole-plan runs only (SyncBatchNorm splits the op at the exchange), not
    // for padded channel counts (the in-kernel form has one channel count) nor for units the reference evaluates twice
    const bool is_conv = o.kind == OP_STEM || o.kind == OP_PW || o.kind == OP_DW || o.kind == OP_KXK;
    const bool fz = is_conv && training && fuse_finalize() && whole && !(o.kind == OP_PW && o.i[2] > 0) &&
                    !((o.kind == OP_PW || o.kind == OP_DW) && (int)o.f[3] > 1);
    const bool h0 = 2 * oi >= sb && 2 * oi < se, h1 = 2 * oi + 1 >= sb && 2 * oi + 1 < se;
    if (!h0 && !h1) continue;
    const bool two_half = o.kind == OP_STEM || o.kind == OP_PW || o.kind == OP_DW || o.kind == OP_KXK || o.kind == OP_CA_MLP ||
                          o.kind == OP_ATT_MLP || o.kind == OP_BN_BWD || o.kind == OP_CA_MLP_BWD || o.kind == OP_ATT_MLP_BWD;
    if (!two_half && !h0) continue;
    const int stage = (h0 && h1) ? 0 : (h0 ? 1 : 2);
    switch (o.kind) {
      case OP_MEMSET: {
        if (hipMemsetAsync(at(ws, o.ws[0]), 0, (size_t)o.ws[1], s) != hipSuccess) {
          lhn_set_error("lhn_plan_run: memset failed");
          rc = 2;
        }
        break;
      }
      case OP_TABLE_FILL: {
        if (skip_tables) break;
        const lhn_buf& b = P->bufs[o.out_buf];
        if (o.i[0])  // deployed conv: (1, bias, slope)
          rc = lhn_table_bias(reinterpret_cast<float*>(at(ws, b.table_off)), b.C, o.out_coff, o.out_C, prm<const float>(params, o.p[0]), o.f[2], stream);
        else
          rc = lhn_table_fill(reinterpret_cast<float*>(at(ws, b.table_off)), b.C, o.out_coff, o.out_C, o.f[0], o.f[1], o.f[2], stream);
        break;
      }
      case OP_STEM: {
        lhn_view y = mkview(P, ws, o.out_buf, o.out_coff, o.out_C);
        const bool bn = conv_has_bn(o);
        lhn_bnfin fin;
        if (bn) {
          fin = mkfin(P, ws, o, params);
          fin.count *= cscale;
        }
        if (h0) rc = lhn_conv_stem_fwd(static_cast<const float*>(io[0]), prm<const float>(params, o.p[0]), &y,
                               (training && o.ws[0] >= 0) ? reinterpret_cast<double*>(at(ws, o.ws[0])) : nullptr, o.i[3], o.i[4], o.i[0], o.i[1],
                               o.i[2], (bn && fz) ? &fin : nullptr, stream);
        if (!rc && bn && h1 && !skip_tables && !deferred && !fz) rc = sep_finalize(fin, reinterpret_cast<const double*>(at(ws, o.ws[0])), training, stream);
        break;
      }
      case OP_PW: {
        lhn_view x = mkview(P, ws, o.in_buf[0], o.in_coff[0], o.in_C[0]);
        lhn_view y;
        float* nchw = nullptr;
        if (o.i[1]) {  // NCHW head: geometry from the input, channels from out_C
          y = x;
          y.data = nullptr; y.table = nullptr; y.gate = nullptr; y.pend = nullptr;
          y.cstride = o.out_C; y.coff = 0; y.C = o.out_C;
          nchw = static_cast<float*>(io[1]);
        } else {
          y = mkview(P, ws, o.out_buf, o.out_coff, o.out_C);
        }
        const bool bn = !o.i[1] && conv_has_bn(o);
        lhn_bnfin fin;
        if (bn) {
          fin = mkfin(P, ws, o, params);
          fin.count *= cscale;
        }
        // i[2], i[3] = real rows / columns of the weight tensor when the views are padded to a multiple of 4 (0 = the views');
        // i[4], i[5] = stack index / number of stacks of an NCHW output [N, S, K, H, W] (hourglassnet.py:136)
        lhn_pw_opts po;
        memset(&po, 0, sizeof(po));
        po.w_rows = o.i[2]; po.w_cols = o.i[3];
        lhn_view extra[2];
        if (o.i[6] > 1) {      // i[6] sources summed on load: in_buf[1..], coefficients f[4..6]
          po.n_extra = o.i[6] - 1;
          for (int e = 0; e < po.n_extra; ++e) {
            extra[e] = mkview(P, ws, o.in_buf[e + 1], o.in_coff[e + 1], o.in_C[e + 1]);
          }
          po.extra = extra;
          for (int e = 0; e < 3; ++e) po.coef[e] = o.f[4 + e];
        }
        // ws[4] >= 0 (plans with a backward): the summed input is also written to the buffer at that byte offset, ws[5] = pixel
        // stride * 65536 + first channel (lhn_pw_opts.sum_out)
        lhn_view sumv;
        if (o.i[6] > 1 && o.ws[4] >= 0) {
          sumv = x;
          sumv.data = reinterpret_cast<float*>(at(ws, o.ws[4]));
          sumv.table = nullptr; sumv.gate = nullptr; sumv.pend = nullptr;
          sumv.cstride = (int)(o.ws[5] >> 16); sumv.coff = (int)(o.ws[5] & 0xffff);
          po.sum_out = &sumv;
        }
        if (nchw && o.i[5] > 1) {
          const int64_t khw = (int64_t)o.out_C * y.H * y.W;
          nchw += (int64_t)o.i[4] * khw;
          po.nchw_batch_stride = (int64_t)o.i[5] * khw;
        }
        if (h0) rc = lhn_conv_pw_fwd2(&x, prm<const float>(params, o.p[0]), bn ? nullptr : prm<const float>(params, o.p[1]), &y,
                              (training && o.ws[0] >= 0) ? reinterpret_cast<double*>(at(ws, o.ws[0])) : nullptr, o.i[0], nchw,
                              (bn && fz) ? &fin : nullptr, &po, stream);
        if (!rc && bn && h1 && !skip_tables && !deferred && !fz)
          rc = sep_finalize(fin, reinterpret_cast<const double*>(at(ws, o.ws[0])), training, stream, o.i[2], (int)o.f[3]);
        break;
      }
      case OP_DW: {
        lhn_view x = mkview(P, ws, o.in_buf[0], o.in_coff[0], o.in_C[0]);
        lhn_view y = mkview(P, ws, o.out_buf, o.out_coff, o.out_C);
        const bool bn = conv_has_bn(o);
        lhn_bnfin fin;
        if (bn) {
          fin = mkfin(P, ws, o, params);
          fin.count *= cscale;
        }
        lhn_view extra;
        const float coef2[2] = {o.f[4], o.f[5]};
        if (o.i[6] > 1) {                                                                 // second source summed on load
          extra = mkview(P, ws, o.in_buf[1], o.in_coff[1], o.in_C[1]);
        }
        lhn_view sumv;           // ws[4], ws[5]: see OP_PW
        const bool so = o.i[6] > 1 && o.ws[4] >= 0;
        if (so) {
          sumv = x;
          sumv.data = reinterpret_cast<float*>(at(ws, o.ws[4]));
          sumv.table = nullptr; sumv.gate = nullptr; sumv.pend = nullptr;
          sumv.cstride = (int)(o.ws[5] >> 16); sumv.coff = (int)(o.ws[5] & 0xffff);
        }
        if (h0) rc = lhn_conv_dw_fwd3(&x, prm<const float>(params, o.p[0]), &y,
                              (training && o.ws[0] >= 0) ? reinterpret_cast<double*>(at(ws, o.ws[0])) : nullptr, o.i[0], o.i[1], o.i[2], o.i[3],
                              (bn && fz) ? &fin : nullptr, o.i[6] > 1 ? &extra : nullptr, coef2,
                              so ? &sumv : nullptr, stream);
        if (!rc && bn && h1 && !skip_tables && !deferred && !fz)
          rc = sep_finalize(fin, reinterpret_cast<const double*>(at(ws, o.ws[0])), training, stream, 0, (int)o.f[3]);
        break;
      }
      case OP_KXK: {
        lhn_view x = mkview(P, ws, o.in_buf[0], o.in_coff[0], o.in_C[0]);
        lhn_view y = mkview(P, ws, o.out_buf, o.out_coff, o.out_C);
        const bool bn = conv_has_bn(o);
        lhn_bnfin fin;
        if (bn) {
          fin = mkfin(P, ws, o, params);
          fin.count *= cscale;
        }
        if (h0) rc = lhn_conv_kxk_fwd(&x, prm<const float>(params, o.p[0]), &y,
                              (training && o.ws[0] >= 0) ? reinterpret_cast<double*>(at(ws, o.ws[0])) : nullptr, o.i[0],
                              (bn && fz) ? &fin : nullptr,
                              o.ws[3] >= 0 ? reinterpret_cast<float*>(at(ws, o.ws[3])) : nullptr, stream);
        if (!rc && bn && h1 && !skip_tables && !deferred && !fz) rc = sep_finalize(fin, reinterpret_cast<const double*>(at(ws, o.ws[0])), training, stream);
        break;
      }
      case OP_FINALIZE: {
        if (skip_tables) break;
        const lhn_buf& b = P->bufs[o.out_buf];
        const int src = o.in_buf[0] >= 0 ? o.in_buf[0] : o.out_buf;   // geometry the statistics were taken over
        const lhn_buf& sb = P->bufs[src];
        rc = lhn_bn_finalize(reinterpret_cast<const double*>(at(ws, o.ws[0])), prm<const float>(params, o.p[0]),
                             prm<const float>(params, o.p[1]), prm<float>(params, o.p[2]), prm<float>(params, o.p[3]),
                             prm<int64_t>(params, o.p[4]), reinterpret_cast<float*>(at(ws, b.table_off)), b.C, o.out_coff,
                             o.out_C, reinterpret_cast<float*>(at(ws, o.ws[1])), (double)sb.N * sb.H * sb.W, o.f[0], o.f[1],
                             o.f[2], training, nullptr, stream);
        break;
      }
      case OP_EW: {
        lhn_view srcs[3];
        for (int k = 0; k < o.i[0]; ++k) {
          srcs[k] = mkview(P, ws, o.in_buf[k], o.in_coff[k], o.in_C[k]);
        }
        lhn_view d = mkview(P, ws, o.out_buf, o.out_coff, o.out_C);
        const float coef[3] = {o.f[4], o.f[5], o.f[6]};
        rc = lhn_ew_fwd3(srcs, o.i[0], o.i[1] ? coef : nullptr, &d, o.f[0], o.i[2], stream);      // i[1]: coefficients given; i[2]: mode
        break;
      }
      case OP_MAXPOOL: {
        lhn_view x = mkview(P, ws, o.in_buf[0], o.in_coff[0], o.in_C[0]);
        lhn_view y = mkview(P, ws, o.out_buf, o.out_coff, o.out_C);
        rc = lhn_maxpool2_fwd(&x, &y, stream);
        break;
      }
      case OP_SHUFFLE: {
        lhn_view a = mkview(P, ws, o.in_buf[0], o.in_coff[0], o.in_C[0]);
        lhn_view b = mkview(P, ws, o.in_buf[1], o.in_coff[1], o.in_C[1]);
        lhn_view d = mkview(P, ws, o.out_buf, o.out_coff, o.out_C);
        rc = lhn_shuffle2_fwd(&a, &b, &d, stream);
        break;
      }
      case OP_AVGPOOL: {
        lhn_view x = mkview(P, ws, o.in_buf[0], o.in_coff[0], o.in_C[0], o.i[2] == 0);
        if (o.ws[1] >= 0 || o.in_buf[1] >= 0) {
          // channel attention: ws[1] = pooling statistics its backward assembles BatchNorm sums from (training plans);
          // in_buf[1] = the pass-through half of a gated unit, copied into the pooled buffer by this launch
          const lhn_bn_slices sl = mkslices(ws, &o.i[5], &o.ws[2], nullptr);
          lhn_view src;
          if (o.in_buf[1] >= 0) src = mkview(P, ws, o.in_buf[1], o.in_coff[1], o.in_C[1]);
          rc = lhn_avgpool_fwd4(&x, reinterpret_cast<float*>(at(ws, o.ws[0])), o.i[0], o.i[1],
                                o.ws[1] >= 0 ? reinterpret_cast<float*>(at(ws, o.ws[1])) : nullptr, &sl, o.in_buf[1] >= 0 ? &src : nullptr, stream);
          break;
        }
        rc = lhn_avgpool_fwd2(&x, reinterpret_cast<float*>(at(ws, o.ws[0])), o.i[0], o.i[1], o.i[3] > 0 ? o.i[3] : x.C, o.i[4], stream);
        break;
      }
      case OP_CA_MLP: {
        const lhn_buf& b = P->bufs[o.out_buf];
        if ((!training || o.ws[3] < 0) && !h0) break;      // not splittable: runs whole on its first half-step
        rc = lhn_ca_mlp_fwd(reinterpret_cast<const float*>(at(ws, o.ws[0])), prm<const float>(params, o.p[0]),
                            prm<const float>(params, o.p[1]), prm<const float>(params, o.p[2]), prm<float>(params, o.p[3]),
                            prm<float>(params, o.p[4]), prm<int64_t>(params, o.p[5]), prm<const float>(params, o.p[6]),
                            prm<const float>(params, o.p[7]), prm<const float>(params, o.p[8]), prm<const float>(params, o.p[9]),
                            (training && o.ws[2] >= 0) ? reinterpret_cast<const float*>(at(ws, o.ws[2])) : nullptr,
                            reinterpret_cast<float*>(at(ws, b.gate_off)), b.C, o.out_coff,
                            reinterpret_cast<float*>(at(ws, o.ws[1])), b.N, o.out_C, o.f[0], o.f[1], training,
                            (training && o.ws[3] >= 0) ? stage : 0, o.ws[3] >= 0 ? reinterpret_cast<double*>(at(ws, o.ws[3])) : nullptr,
                            cscale, stream);
        break;
      }
      case OP_ATT_MLP: {  // p: gamma, beta, rmean, rvar, nbt, w3, b3, wl, bl; ws: pooled, save, mask
        const lhn_buf& b = P->bufs[o.out_buf];
        if ((!training || o.ws[3] < 0) && !h0) break;
        rc = lhn_att_mlp_fwd(reinterpret_cast<const float*>(at(ws, o.ws[0])), prm<const float>(params, o.p[0]),
                             prm<const float>(params, o.p[1]), prm<float>(params, o.p[2]), prm<float>(params, o.p[3]),
                             prm<int64_t>(params, o.p[4]), prm<const float>(params, o.p[5]), prm<const float>(params, o.p[6]),
                             prm<const float>(params, o.p[7]), prm<const float>(params, o.p[8]),
                             (training && o.ws[2] >= 0) ? reinterpret_cast<const float*>(at(ws, o.ws[2])) : nullptr,
                             reinterpret_cast<float*>(at(ws, b.gate_off)), b.C, o.out_coff,
                             reinterpret_cast<float*>(at(ws, o.ws[1])), b.N, o.out_C, o.f[0], o.f[1], training,
                             (training && o.ws[3] >= 0) ? stage : 0, o.ws[3] >= 0 ? reinterpret_cast<double*>(at(ws, o.ws[3])) : nullptr,
                             cscale, stream);
        break;
      }
      case OP_SE_MLP: {  // p: w1, b1, w2, b2; ws: pooled, save; i[0] = J
        const lhn_buf& b = P->bufs[o.out_buf];
        rc = lhn_se_mlp_fwd2(reinterpret_cast<const float*>(at(ws, o.ws[0])), prm<const float>(params, o.p[0]),
                             prm<const float>(params, o.p[1]), prm<const float>(params, o.p[2]), prm<const float>(params, o.p[3]),
                             reinterpret_cast<float*>(at(ws, b.gate_off)), b.C, o.out_coff, reinterpret_cast<float*>(at(ws, o.ws[1])),
                             b.N, o.out_C, o.i[0], o.i[1], stream);
        break;
      }
      // ------------------------------------------------------------------ backward
      case OP_STEM_BWD: {
        lhn_view y = mkview(P, ws, o.out_buf, o.out_coff, o.out_C);
        lhn_gradview g = mkgrad(P, ws, o.out_buf, o.i[5] != 0);
        rc = lhn_conv_stem_bwd(static_cast<const float*>(io[0]), &y, &g, prm<float>(grads, o.p[1]), o.i[3], o.i[4], o.i[0],
                               o.i[1], o.i[2], nrep, rstr, stream);
        break;
      }
      case OP_PW_BWD: {
        lhn_view x = mkview(P, ws, o.in_buf[0], o.in_coff[0], o.in_C[0]);
        lhn_view y;
        lhn_gradview g;
        const float* nchw = nullptr;
        if (o.i[1]) {
          y = x;
          y.data = nullptr; y.table = nullptr; y.gate = nullptr;
          y.cstride = o.out_C; y.coff = 0; y.C = o.out_C;
          g.dz = nullptr; g.dpool = nullptr; g.coef = nullptr;
          nchw = static_cast<const float*>(io[1]);
        } else {
          y = mkview(P, ws, o.out_buf, o.out_coff, o.out_C);
          g = mkgrad(P, ws, o.out_buf, o.i[5] != 0);
        }
        float* dx = o.i[2] ? reinterpret_cast<float*>(at(ws, P->bufs[o.in_buf[0]].grad_off)) : nullptr;
        lhn_pw_opts po;     // i[3], i[4] = weight rows / columns; i[6], i[7] = stack index / stacks (see OP_PW)
        memset(&po, 0, sizeof(po));
        po.w_rows = o.i[3]; po.w_cols = o.i[4];
        if (nchw && o.i[7] > 1) {
          const int64_t khw = (int64_t)o.out_C * y.H * y.W;
          nchw += (int64_t)o.i[6] * khw;
          po.nchw_batch_stride = (int64_t)o.i[7] * khw;
        }
        const lhn_bnsum bs = mkbns(ws, o.ws[0], o.ws[1], (int)o.f[6], (int)o.f[7]);      // ws[0..1], f[6..7]: the input's producer
        rc = lhn_conv_pw_bwd3(&x, prm<const float>(params, o.p[0]), &y, &g, dx, o.i[2] == 2, prm<float>(grads, o.p[1]),
                              prm<float>(grads, o.p[2]), o.i[0], nchw, nrep, rstr, &po, bs.sums ? &bs : nullptr, stream);
        break;
      }
      case OP_DW_BWD: {
        lhn_view x = mkview(P, ws, o.in_buf[0], o.in_coff[0], o.in_C[0]);
        lhn_view y = mkview(P, ws, o.out_buf, o.out_coff, o.out_C);
        lhn_gradview g = mkgrad(P, ws, o.out_buf, o.i[5] != 0);
        float* dx = o.i[4] ? reinterpret_cast<float*>(at(ws, P->bufs[o.in_buf[0]].grad_off)) : nullptr;
        // ws[4], ws[5] >= 0: BatchNorm-backward sums / saved statistics of the convolution that produced x, i[6] its channel
        // count, i[7] the producer channel of x's first channel (see lhn_conv_dw_bwd2)
        // ws[2], ws[3] >= 0: gradients of residual sums that read x, added to the stored dx (lhn_conv_dw_bwd3)
        if (o.ws[2] >= 0 || o.ws[3] >= 0) {
          rc = lhn_conv_dw_bwd3(&x, prm<const float>(params, o.p[0]), &y, &g, dx, o.i[4] == 2, prm<float>(grads, o.p[1]), o.i[0], o.i[1],
                                o.i[2], o.i[3], nrep, rstr, o.ws[2] >= 0 ? reinterpret_cast<const float*>(at(ws, o.ws[2])) : nullptr,
                                o.ws[3] >= 0 ? reinterpret_cast<const float*>(at(ws, o.ws[3])) : nullptr, stream);
          break;
        }
        rc = lhn_conv_dw_bwd2(&x, prm<const float>(params, o.p[0]), &y, &g, dx, o.i[4] == 2, prm<float>(grads, o.p[1]), o.i[0],
                              o.i[1], o.i[2], o.i[3], nrep, rstr, o.ws[4] >= 0 ? reinterpret_cast<double*>(at(ws, o.ws[4])) : nullptr,
                              o.ws[5] >= 0 ? reinterpret_cast<const float*>(at(ws, o.ws[5])) : nullptr, o.i[6], o.i[7], stream);
        break;
      }
      case OP_KXK_BWD: {
        lhn_view x = mkview(P, ws, o.in_buf[0], o.in_coff[0], o.in_C[0]);
        lhn_view y = mkview(P, ws, o.out_buf, o.out_coff, o.out_C);
        lhn_gradview g = mkgrad(P, ws, o.out_buf, o.i[5] != 0);
        float* dx = o.i[2] ? reinterpret_cast<float*>(at(ws, P->bufs[o.in_buf[0]].grad_off)) : nullptr;
        rc = lhn_conv_kxk_bwd(&x, prm<const float>(params, o.p[0]), &y, &g, dx, o.i[2] == 2, prm<float>(grads, o.p[1]), o.i[0],
                              nrep, rstr, o.ws[3] >= 0 ? reinterpret_cast<float*>(at(ws, o.ws[3])) : nullptr, stream);
        break;
      }
      case OP_BN_BWD: {
        const lhn_buf& b = P->bufs[o.out_buf];
        lhn_view y = mkview(P, ws, o.out_buf, o.out_coff, o.out_C);
        lhn_gradview g = mkgrad(P, ws, o.out_buf, false);
        double* sums = reinterpret_cast<double*>(at(ws, o.ws[0]));
        const float* save = reinterpret_cast<const float*>(at(ws, o.ws[1]));
        lhn_bnbwdfin fin;
        fin.counter = reinterpret_cast<uint32_t*>(at(ws, o.ws[2]));
        fin.gamma = prm<const float>(params, o.p[0]);
        fin.coef = reinterpret_cast<float*>(at(ws, b.coef_off));
        fin.dgamma = prm<float>(grads, o.p[1]);
        fin.dbeta = prm<float>(grads, o.p[2]);
        fin.count = (double)b.N * b.H * b.W * cscale;
        fin.cstride = b.C; fin.coff = o.out_coff; fin.C = o.out_C;
        if (fin.counter && fuse_finalize() && whole && !o.i[1]) {
          rc = lhn_bn_bwd_reduce(&y, &g, save, sums, &fin, stream);
        } else {
          if (h0 && !o.i[1]) rc = lhn_bn_bwd_reduce(&y, &g, save, sums, nullptr, stream);      // i[1]: sums come from the reader's backward
          if (!rc && h1)
            rc = lhn_bn_bwd_finalize2(sums, fin.gamma, save, fin.coef, b.C, o.out_coff, o.i[0] > 0 ? o.i[0] : o.out_C, o.out_C, fin.count,
                                      fin.dgamma, fin.dbeta, pscale, stream);      // i[0]: real channels of a padded output
        }
        break;
      }
      case OP_EW_BWD: {
        lhn_view src = mkview(P, ws, o.in_buf[0], o.in_coff[0], o.in_C[0]);
        lhn_view d = mkview(P, ws, o.out_buf, o.out_coff, o.out_C);
        const lhn_buf& db = P->bufs[o.out_buf];
        if (o.i[1] == 1) {            // product: in_buf[1] = the other operand
          lhn_view other = mkview(P, ws, o.in_buf[1], o.in_coff[1], o.in_C[1]);
          rc = lhn_ew_mul_bwd(&src, &other, &d, reinterpret_cast<const float*>(at(ws, db.grad_off)),
                              reinterpret_cast<float*>(at(ws, P->bufs[o.in_buf[0]].grad_off)), o.i[0], stream);
        } else if (o.i[1] == 2) {     // bilinearly resampled source
          rc = lhn_bilinear_bwd(&src, &d, reinterpret_cast<const float*>(at(ws, db.grad_off)),
                                reinterpret_cast<float*>(at(ws, P->bufs[o.in_buf[0]].grad_off)), o.i[0], o.f[0], stream);
        } else if (o.i[1] == 3) {     // 2 or 3 sources of the destination's resolution in one pass: in_buf[0..2]; accumulate flags i[0], i[2],
                                      // i[3]; their producers' sums ws[0..5], (C, coff) in i[4..7] and f[4..5]
          lhn_view srcs[3];
          float* dsrcs[3];
          int acc[3] = {o.i[0], o.i[2], o.i[3]};
          lhn_bnsum bsv[3] = {mkbns(ws, o.ws[0], o.ws[1], o.i[4], o.i[5]), mkbns(ws, o.ws[2], o.ws[3], o.i[6], o.i[7]),
                              mkbns(ws, o.ws[4], o.ws[5], (int)o.f[4], (int)o.f[5])};
          const lhn_bnsum* bsp[3];
          int ns = 0;
          for (; ns < 3 && o.in_buf[ns] >= 0; ++ns) {
            srcs[ns] = mkview(P, ws, o.in_buf[ns], o.in_coff[ns], o.in_C[ns]);
            dsrcs[ns] = reinterpret_cast<float*>(at(ws, P->bufs[o.in_buf[ns]].grad_off));
            bsp[ns] = bsv[ns].sums ? &bsv[ns] : nullptr;
          }
          rc = lhn_ew_bwd_multi(srcs, ns, &d, reinterpret_cast<const float*>(at(ws, db.grad_off)),
                                reinterpret_cast<const float*>(at(ws, db.dpool_off)), o.f[0], dsrcs, acc, bsp, stream);
        } else {
          const lhn_bnsum bs = mkbns(ws, o.ws[0], o.ws[1], o.i[4], o.i[5]);
          rc = lhn_ew_bwd3(&src, &d, reinterpret_cast<const float*>(at(ws, db.grad_off)),
                           reinterpret_cast<const float*>(at(ws, db.dpool_off)), o.f[0],
                           reinterpret_cast<float*>(at(ws, P->bufs[o.in_buf[0]].grad_off)), o.i[0], bs.sums ? &bs : nullptr, stream);
        }
        break;
      }
      case OP_MAXPOOL_BWD: {
        lhn_view x = mkview(P, ws, o.in_buf[0], o.in_coff[0], o.in_C[0]);
        lhn_view y = mkview(P, ws, o.out_buf, o.out_coff, o.out_C);
        const lhn_bnsum bs = mkbns(ws, o.ws[0], o.ws[1], o.i[4], o.i[5]);
        // ws[2] / i[1], i[2]: gradient of a plain sum that also reads x (base, pixel stride, first channel); ws[3] / i[3], i[6], i[7]:
        // gradient of an adaptive average pool of x (base, pixel stride, first channel, OH << 16 | OW) -- lhn_grad_adds
        lhn_grad_adds ad;
        memset(&ad, 0, sizeof(ad));
        if (o.ws[2] >= 0) {
          ad.same = reinterpret_cast<const float*>(at(ws, o.ws[2]));
          ad.same_cstride = o.i[1]; ad.same_coff = o.i[2];
        }
        if (o.ws[3] >= 0) {
          ad.pooled = reinterpret_cast<const float*>(at(ws, o.ws[3]));
          ad.pooled_cstride = o.i[3]; ad.pooled_coff = o.i[6];
          ad.OH = o.i[7] >> 16; ad.OW = o.i[7] & 0xffff;
        }
        rc = lhn_maxpool2_bwd3(&x, &y, reinterpret_cast<const float*>(at(ws, P->bufs[o.out_buf].grad_off)),
                               reinterpret_cast<float*>(at(ws, P->bufs[o.in_buf[0]].grad_off)), o.i[0], bs.sums ? &bs : nullptr,
                               (ad.same || ad.pooled) ? &ad : nullptr, stream);
        break;
      }
      case OP_AVGPOOL_BWD: {
        lhn_view x = mkview(P, ws, o.in_buf[0], o.in_coff[0], o.in_C[0]);
        const lhn_bnsum bs = mkbns(ws, o.ws[1], o.ws[2], o.i[5], o.i[6]);
        rc = lhn_avgpool_bwd3(&x, reinterpret_cast<const float*>(at(ws, o.ws[0])), o.i[0], o.i[1], o.i[3] > 0 ? o.i[3] : x.C, o.i[4],
                              reinterpret_cast<float*>(at(ws, P->bufs[o.in_buf[0]].grad_off)), o.i[2], bs.sums ? &bs : nullptr, stream);
        break;
      }
      case OP_SHUFFLE_BWD: {     // i[0], i[1]: 0 = no gradient wanted, 1 = store, 2 = accumulate (operand a, b)
        lhn_view a = mkview(P, ws, o.in_buf[0], o.in_coff[0], o.in_C[0]);
        lhn_view b = mkview(P, ws, o.in_buf[1], o.in_coff[1], o.in_C[1]);
        lhn_view d = mkview(P, ws, o.out_buf, o.out_coff, o.out_C);
        rc = lhn_shuffle2_bwd(&a, &b, &d, reinterpret_cast<const float*>(at(ws, P->bufs[o.out_buf].grad_off)),
                              o.i[0] ? reinterpret_cast<float*>(at(ws, P->bufs[o.in_buf[0]].grad_off)) : nullptr, o.i[0] == 2,
                              o.i[1] ? reinterpret_cast<float*>(at(ws, P->bufs[o.in_buf[1]].grad_off)) : nullptr, o.i[1] == 2, stream);
        break;
      }
      case OP_GATE_REDUCE: {
        lhn_view y = mkview(P, ws, o.out_buf, o.out_coff, o.out_C, false);
        float* dg = reinterpret_cast<float*>(at(ws, o.ws[3]));
        const lhn_bn_slices sl = mkslices(ws, &o.i[0], &o.ws[5], nullptr);
        // (dgate lives in the arena the backward zeroes with one memset: prezeroed)
        rc = lhn_gate_bwd_reduce3(&y, reinterpret_cast<const float*>(at(ws, P->bufs[o.out_buf].grad_off)), dg,
                                  o.ws[4] >= 0 ? dg + (size_t)y.N * y.C : nullptr, &sl, 1, stream);
        break;
      }
      case OP_CA_MLP_BWD: {
        const lhn_buf& b = P->bufs[o.out_buf];
        if (o.ws[4] < 0 && !h0) break;
        const lhn_bn_slices sl = mkslices(ws, &o.i[0], &o.ws[6], &o.ws[8]);
        const float* dgp = reinterpret_cast<const float*>(at(ws, o.ws[3]));
        rc = lhn_ca_mlp_bwd2(reinterpret_cast<const float*>(at(ws, o.ws[0])), prm<const float>(params, o.p[0]),
                            prm<const float>(params, o.p[1]), prm<const float>(params, o.p[2]), prm<const float>(params, o.p[3]),
                            o.ws[2] >= 0 ? reinterpret_cast<const float*>(at(ws, o.ws[2])) : nullptr,
                            reinterpret_cast<const float*>(at(ws, o.ws[1])), reinterpret_cast<const float*>(at(ws, o.ws[3])),
                            reinterpret_cast<float*>(at(ws, b.dpool_off)), b.C, o.out_coff, b.H, b.W, prm<float>(grads, o.p[4]),
                            prm<float>(grads, o.p[5]), prm<float>(grads, o.p[6]), prm<float>(grads, o.p[7]),
                            prm<float>(grads, o.p[8]), prm<float>(grads, o.p[9]), prm<float>(grads, o.p[10]), b.N, o.out_C,
                            o.ws[4] >= 0 ? stage : 0, o.ws[4] >= 0 ? reinterpret_cast<double*>(at(ws, o.ws[4])) : nullptr, cscale, pscale,
                            o.ws[5] >= 0 ? dgp + (size_t)b.N * b.C : nullptr, o.ws[5] >= 0 ? reinterpret_cast<const float*>(at(ws, o.ws[5])) : nullptr,
                            &sl, stream);
        break;
      }
      case OP_ATT_MLP_BWD: {  // p: gamma, beta, w3, wl (params) | dgamma, dbeta, dw3, db3, dwl, dbl (grads); ws: pooled, save, mask, dgate
        const lhn_buf& b = P->bufs[o.out_buf];
        if (o.ws[4] < 0 && !h0) break;
        rc = lhn_att_mlp_bwd(reinterpret_cast<const float*>(at(ws, o.ws[0])), prm<const float>(params, o.p[0]),
                             prm<const float>(params, o.p[1]), prm<const float>(params, o.p[2]), prm<const float>(params, o.p[3]),
                             o.ws[2] >= 0 ? reinterpret_cast<const float*>(at(ws, o.ws[2])) : nullptr,
                             reinterpret_cast<float*>(at(ws, o.ws[1])), reinterpret_cast<const float*>(at(ws, o.ws[3])),
                             reinterpret_cast<float*>(at(ws, b.dpool_off)), b.C, o.out_coff, b.H, b.W, prm<float>(grads, o.p[4]),
                             prm<float>(grads, o.p[5]), prm<float>(grads, o.p[6]), prm<float>(grads, o.p[7]),
                             prm<float>(grads, o.p[8]), prm<float>(grads, o.p[9]), b.N, o.out_C,
                             o.ws[4] >= 0 ? stage : 0, o.ws[4] >= 0 ? reinterpret_cast<double*>(at(ws, o.ws[4])) : nullptr, cscale, pscale,
                             stream);
        break;
      }
      case OP_SE_MLP_BWD: {  // p: w1, w2 (params) | dw1, db1, dw2, db2 (grads); ws: pooled, save, -, dgate; i[0] = J
        const lhn_buf& b = P->bufs[o.out_buf];
        rc = lhn_se_mlp_bwd2(reinterpret_cast<const float*>(at(ws, o.ws[0])), prm<const float>(params, o.p[0]),
                             prm<const float>(params, o.p[1]), reinterpret_cast<const float*>(at(ws, o.ws[1])),
                             reinterpret_cast<const float*>(at(ws, o.ws[3])), reinterpret_cast<float*>(at(ws, b.dpool_off)), b.C,
                             o.out_coff, b.H, b.W, prm<float>(grads, o.p[2]), prm<float>(grads, o.p[3]), prm<float>(grads, o.p[4]),
                             prm<float>(grads, o.p[5]), b.N, o.out_C, o.i[0], o.i[1], stream);
        break;
      }
      default:
        lhn_set_error("lhn_plan_run: unknown op kind %d at index %zu", o.kind, oi);
        rc = 1;
    }
  }
  return rc;
}

// SyncBatchNorm driver entry: run the half-steps [step_begin, step_end) of a phase (see run_ops).  The caller all-reduces
// the statistics buffer of op i between step 2*i and step 2*i+1.
int lhn_plan_run_range(void* plan, int phase, int64_t step_begin, int64_t step_end, void* ws, void* const* params,
                       void* const* grads, void* const* io, int training, int grad_replicas, int64_t grad_rep_stride,
                       double count_scale, float pgrad_scale, void* stream) {
  LHN_CHECK_ARG(plan && ws && params && io && step_begin >= 0 && step_end >= step_begin && count_scale >= 1, "lhn_plan_run_range: bad argument");
  LHN_CHECK_ARG(phase == 0 || (phase == 1 && grads), "lhn_plan_run_range: phase %d", phase);
  return run_ops(static_cast<const Plan*>(plan), phase, ws, params, grads, io, training, grad_replicas < 1 ? 1 : grad_replicas,
                 grad_rep_stride, stream, (size_t)step_begin, (size_t)step_end, count_scale, pgrad_scale);
}

static bool graphs_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("LHN_GRAPH");
    v = (e && e[0] == '1') ? 1 : 0;
  }
  return v == 1;
}
static uint64_t hash_ptrs(void* const* a, const std::vector<lhn_op>& ops, bool grads) {
  // only the entries the ops reference (p[] indices) matter; FNV-1a over their values
  uint64_t h = 1469598103934665603ull;
  if (!a) return h;
  for (const lhn_op& o : ops)
    for (int k = 0; k < 12; ++k)
      if (o.p[k] >= 0) {
        h ^= reinterpret_cast<uint64_t>(a[o.p[k]]);
        h *= 1099511628211ull;
      }
  (void)grads;
  return h;
}

// Runs one phase.  With LHN_RUN_GRAPH in `training` (or LHN_GRAPH=1 in the environment) the launch sequence of a (phase, pointer set) is captured into a hipGraph the second
// time it is seen and replayed afterwards (one graph launch instead of 150-400 kernel launches: the small-batch /
// low-resolution end of the network is launch-bound).  The plan is static, so a graph is valid for as long as the
// workspace, parameter, gradient and io pointers repeat; a different pointer set simply gets its own graph (LRU of 8),
// and a plan whose pointers never repeat falls back to plain launches.
int lhn_plan_run(void* plan, int phase, void* ws, void* const* params, void* const* grads, void* const* io, int training,
                 int grad_replicas, int64_t grad_rep_stride, void* stream) {
  const int nrep = grad_replicas < 1 ? 1 : grad_replicas;
  const int64_t rstr = grad_rep_stride;
  LHN_CHECK_ARG(plan && ws && params && io, "lhn_plan_run: null argument");
  LHN_CHECK_ARG(phase == 0 || (phase == 1 && grads), "lhn_plan_run: phase %d", phase);
  Plan* P = static_cast<Plan*>(plan);
  // (the legacy default stream cannot be captured: graphs need the caller to run on a real stream)
  const bool want_graph = graphs_enabled() || (training & LHN_RUN_GRAPH) != 0;
  training &= ~LHN_RUN_GRAPH;
  if (!want_graph || stream == nullptr || P->graph_misses > 64) return run_ops(P, phase, ws, params, grads, io, training, nrep, rstr, stream);
  const std::vector<lhn_op>& ops = phase == 0 ? P->fwd : P->bwd;
  const uint64_t ph = hash_ptrs(params, ops, false), gh = phase == 1 ? hash_ptrs(grads, ops, true) : 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  GraphEntry* e = nullptr;
  for (GraphEntry& g : P->graphs)
    if (g.phase == phase && g.training == training && g.nrep == nrep && g.rstr == rstr && g.ws == ws && g.io0 == io[0] &&
        g.io1 == io[1] && g.phash == ph && g.ghash == gh) {
      e = &g;
      break;
    }
  ++P->tick;
  if (e && e->exec) {
    e->last_use = P->tick;
    if (hipGraphLaunch(e->exec, s) != hipSuccess) {
      lhn_set_error("lhn_plan_run: hipGraphLaunch failed");
      return 2;
    }
    return 0;
  }
  if (!e) {  // first sighting: run eagerly (also initialises every kernel's one-time attributes), remember the key
    ++P->graph_misses;
    if (P->graphs.size() >= 8) {
      size_t victim = 0;
      for (size_t i = 1; i < P->graphs.size(); ++i)
        if (P->graphs[i].last_use < P->graphs[victim].last_use) victim = i;
      if (P->graphs[victim].exec) (void)hipGraphExecDestroy(P->graphs[victim].exec);
      P->graphs.erase(P->graphs.begin() + victim);
    }
    P->graphs.push_back(GraphEntry{phase, training, nrep, 1, rstr, ws, io[0], io[1], ph, gh, nullptr, P->tick});
    return run_ops(P, phase, ws, params, grads, io, training, nrep, rstr, stream);
  }
  // second sighting: capture, instantiate, launch
  e->last_use = P->tick;
  hipGraph_t graph = nullptr;
  if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) != hipSuccess)
    return run_ops(P, phase, ws, params, grads, io, training, nrep, rstr, stream);
  const int rc = run_ops(P, phase, ws, params, grads, io, training, nrep, rstr, stream);
  const hipError_t ec = hipStreamEndCapture(s, &graph);
  if (rc != 0 || ec != hipSuccess || !graph) {
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipGetLastError();
    P->graph_misses = 1000;     // capture is not possible here: stay on plain launches
    return rc != 0 ? rc : run_ops(P, phase, ws, params, grads, io, training, nrep, rstr, stream);
  }
  hipGraphExec_t exec = nullptr;
  const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (ei != hipSuccess || !exec) {
    (void)hipGetLastError();
    P->graph_misses = 1000;
    return run_ops(P, phase, ws, params, grads, io, training, nrep, rstr, stream);
  }
  e->exec = exec;
  P->graph_misses = 0;
  if (hipGraphLaunch(exec, s) != hipSuccess) {
    lhn_set_error("lhn_plan_run: hipGraphLaunch failed");
    return 2;
  }
  return 0;
}
}  // extern "C"
