"""CPU: host logic of the plan compiler (litehandnet_amd/plan.py) -- no GPU, no kernel launch.  The module mirrors emit
their launch records into a PlanBuilder; these tests check the structure the C executor relies on and tie bench.py's
algorithmic-byte constants to the plan that is actually run."""
import importlib.util
import os

import pytest

from litehandnet_amd import get_model
from litehandnet_amd.config import litehandnet_cfg
from litehandnet_amd.plan import AVGPOOL, DW, EW, KXK, MAXPOOL, PW, STEM, PlanBuilder

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(variant, n=2, size=256, backward=True, **kw):
    cfg = litehandnet_cfg(variant, image_size=size, **kw)
    cfg.MODEL["ca_dropout"] = 0.0
    m = get_model(cfg)
    tensors = list(m.state_dict(keep_vars=True).values())
    pb = PlanBuilder(n, {id(t): j for j, t in enumerate(tensors)}, image_hw=(size, size), with_backward=backward, p_drop=0.0)
    y = m.emit(pb, pb.image())
    if y.buf != -2:
        pb.set_output(y)
    return m, pb, y


def _conv_bytes(pb):
    return sum(4 * (r["x"].H * r["x"].W * r["x"].C + r["out"].H * r["out"].W * r["out"].C)
               for r in pb.recs if r["op"] in (STEM, PW, DW, KXK))


@pytest.mark.parametrize("variant", ["A", "B", "M"])
def test_plan_structure(variant):
    m, pb, y = _build(variant)
    assert y.buf == -2 and (y.C, y.H, y.W) == (21, 64, 64)            # NCHW head: [N,21,64,64] written straight to the caller
    # every launch reads buffers that an earlier launch (or the image) wrote
    written = {-1}
    for r in pb.recs:
        ins = [r["x"]] if r["op"] in (STEM, PW, DW, KXK, MAXPOOL, AVGPOOL) else (r["srcs"] if r["op"] == EW else [])
        for t in ins:
            assert t.buf in written, (r["op"], t.buf)
        if "out" in r and r["out"] is not None:
            written.add(r["out"].buf)
    cb, cf, cbw, nf, nb = pb.finalize()
    assert nf >= len(pb.recs) and nb > nf                               # backward has at least one launch per forward record
    assert pb.grad_aliases >= 4                                         # residual-add sources share their consumer's gradient
    # SyncBatchNorm cut points: one per statistics buffer and direction, in launch order
    for phase in (0, 1):
        ois = [oi for oi, _, _, _ in pb.sync_points[phase]]
        assert ois == sorted(ois) and len(ois) >= 60
    # channel slices stay inside their buffers and 16-byte aligned
    for b in pb.bufs:
        assert b.C % 4 == 0 and b.off["data"] % 16 == 0 and b.off["table"] % 16 == 0
    assert pb.total_bytes < 2 * 1024 ** 3                                # N=2 workspace; scales linearly with N


def test_algorithmic_bytes_match_bench_constants():
    """bench.py prices the roofline with SURVEY section 8d's algorithmic bytes per image (4 B x conv in+out elements).  The
    plan of variant B moves exactly those convolutions; A and M add BatchNorm-only statistics passes on top."""
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n = 2
    for variant, lo, hi in (("B", 0.995, 1.005), ("A", 1.0, 1.25), ("M", 1.0, 1.25)):
        _, pb, _ = _build(variant, n=n, backward=False)
        per_img = _conv_bytes(pb)                 # record geometry is per image
        ratio = per_img / bench.ALG_FWD_BYTES[variant]
        assert lo <= ratio <= hi, (variant, per_img, bench.ALG_FWD_BYTES[variant])
    assert bench.LOSS_BYTES == 12 * 21 * 64 * 64 and bench.HBM_PEAK_GBS == 8000.0


def test_plan_shapes_and_options():
    # 224x224 (freihand configs): odd 7x7 lowest level, ceil-mode pooling
    _, pb, y = _build("B", size=224)
    assert (y.H, y.W) == (56, 56)
    assert min(r["out"].H for r in pb.recs if r["op"] in (PW, DW)) == 7
    # inference plan: no backward launches, smaller arena
    _, pf, _ = _build("B", backward=False)
    _, pt, _ = _build("B", backward=True)
    nb_f, nb_t = pf.finalize()[4], pt.finalize()[4]
    assert nb_f == 0 and nb_t > 0 and pf.total_bytes < pt.total_bytes
    # squeeze-and-excitation and SiLU variants compile too
    _build("B", msrb_ca="se", rbu_ca="se")
    _, ps, _ = _build("A", activation="silu")
    assert sum(1 for r in ps.recs if r["op"] == EW and r["slope"] == 2.0) > 20      # SiLU lives in elementwise combines


def test_eval_table_reuse_signature():
    """CompiledPlan._tables_current: reuse only while neither torch nor our train-mode kernels touched a parameter."""
    import torch

    from litehandnet_amd import plan as planmod
    p = object.__new__(planmod.CompiledPlan)            # no GPU: exercise the bookkeeping only
    w, rm = torch.nn.Parameter(torch.zeros(4)), torch.zeros(4)
    p.state_tensors, p._table_sig, p.handle = [w, rm], None, None
    assert not p._tables_current()                       # first eval run builds the tables
    assert p._tables_current()                           # nothing changed
    rm.add_(1.0)                                         # load_state_dict / in-place edit: version counter moves
    assert not p._tables_current() and p._tables_current()
    with torch.no_grad():
        w.mul_(2.0)                                      # optimizer step
    assert not p._tables_current() and p._tables_current()
    planmod._TRAIN_RUNS += 1                             # any train-mode forward in the process (raw-pointer updates)
    assert not p._tables_current() and p._tables_current()
    # `.data` writes bump no version counter (EMA / weight surgery): invalidate_tables() is the documented way
    w.data.mul_(3.0)
    assert p._tables_current()                           # invisible ...
    import litehandnet_amd
    litehandnet_amd.invalidate_tables()
    assert not p._tables_current() and p._tables_current()
    # re-homing a LATE tensor (FlatParams: p.data = view of a flat buffer) is seen through its data pointer
    many = [torch.nn.Parameter(torch.zeros(4)) for _ in range(8)]
    p.state_tensors, p._table_sig = many, None
    assert not p._tables_current() and p._tables_current()
    many[6].data = torch.zeros(4)
    assert not p._tables_current() and p._tables_current()
    with torch.inference_mode():
        p.state_tensors = [torch.zeros(4)]               # no version counter: never reuse
    assert not p._tables_current() and not p._tables_current()


def test_two_part_pass_through():
    """RepBasicUnit without attention returns cat(left, branch) as a two-part tensor: no copy record for the pass-through
    half, whole-tensor consumers run once per part, and every part of the decoder's adds keeps its own gradient range."""
    from litehandnet_amd.plan import MAXPOOL
    _, pb, _ = _build("B", backward=True)
    copies = [r for r in pb.recs if r["op"] == EW and len(r["srcs"]) == 1 and r["slope"] == 1.0]
    assert len(copies) == 4                               # the four GATED units (stem, neck) still copy their left half
    assert sum(1 for r in pb.recs if r["op"] == MAXPOOL) == 1 + 2 * 3     # stem pool + three two-part encoder pools
    pb.finalize()
    assert not pb._needs_zero_grad


def test_lazy_sums_are_summed_on_load(monkeypatch):
    """MSRB's residual sums (litehourglass.py:41-49) and MSAB's `m + x` (liteHandNet.py:164) are not launched in forward:
    their readers list the operands (2 for the depthwise 3x3, 3 with a doubled coefficient for the closing 1x1).  In a plan
    with a backward the readers also write the sum (lhn_pw_opts.sum_out; their slices tile the buffer) so the backward has
    no combine to launch; with LHN_SUM_OUT=0 the backward list materialises each sum exactly once."""
    from litehandnet_amd.plan import DW, EW, PW
    _, pb, _ = _build("B", backward=True)
    cb, cf, cbw, nf, nb = pb.finalize()
    fwd = [cf[i] for i in range(nf)]
    writers = [o for o in fwd if o.kind in (DW, PW) and o.i[6] > 1]
    assert len(writers) == 6 and all(o.ws[4] >= 0 for o in writers) and len(pb._sum_written) == 4
    datas = {cb[j].data_off: j for j in pb._sum_written}
    for o in writers:
        j = datas[o.ws[4]]
        assert o.ws[5] >> 16 == cb[j].C and (o.ws[5] & 0xffff) == o.in_coff[0]
    assert not [cbw[i] for i in range(nb) if cbw[i].kind == EW]
    assert not [o for o in fwd if o.ws[4] >= 0 and o.kind in (DW, PW) and o.i[6] <= 1]
    monkeypatch.setenv("LHN_SUM_OUT", "0")
    _, pb, _ = _build("B", backward=True)
    lazy = [r for r in pb.recs if r["op"] == EW and r.get("lazy")]
    assert len(lazy) == 4                                           # 2 MSRBs x 2 rounds
    cb, cf, cbw, nf, nb = pb.finalize()
    fwd = [cf[i] for i in range(nf)]
    # (launched combines: every record that is neither summed on load nor -- the pass-through copy of a gated RepBasicUnit --
    # done by the attention's pooling launch)
    assert sum(1 for o in fwd if o.kind == EW) == sum(1 for r in pb.recs if r["op"] == EW and not r.get("lazy") and not r.get("fwd_fused"))
    assert sum(1 for r in pb.recs if r.get("fwd_fused")) == 4 == sum(1 for o in fwd if o.kind == AVGPOOL and o.in_buf[1] >= 0)
    two = [o for o in fwd if o.kind == DW and o.i[6] == 2]
    three = [o for o in fwd if o.kind == PW and o.i[6] == 3]
    assert len(two) == 4 and len(three) == 2
    for o in three:
        assert sorted(o.f[4:7]) == [1.0, 1.0, 2.0]                  # out + ca(cat2) + x with out = x + ca(cat1)
    bwd = [cbw[i] for i in range(nb)]
    mats = [o for o in bwd if o.kind == EW]
    assert len(mats) == 4 and all(o.i[1] == 1 for o in mats)
    # inference plans do not even allocate the sums
    _, pf, _ = _build("B", backward=False)
    pf.finalize()
    lz = [b for b in pf.bufs if b.lazy is not None]
    assert len(lz) == 4 and len({b.off["data"] for b in lz} | {pf.act_bytes}) <= 5
    # variant A: the sum in front of MSAB's closing 1x1
    _, pa, _ = _build("A", backward=False)
    assert sum(1 for r in pa.recs if r["op"] == EW and r.get("lazy")) == 2


def test_bn_backward_sums_ride_in_the_reader(monkeypatch):
    """RepBasicUnit's 1x1 -> 3x3 depthwise (and the stem's 3x3 -> 3x3 depthwise): the depthwise backward is the only reader of
    the producer's output, so it also accumulates that BatchNorm's backward sums (lhn_conv_dw_bwd2) and the producer's BN_BWD
    op skips its reduce pass: 19 of variant B's 52.  The 12 convolutions that write GATED buffers (8 MSRB branches, 4 gated
    RepBasicUnits) get their sums from the attention's backward (lhn_ca_mlp_bwd2: lhn_bn_slices) -- also no reduce pass.
    LHN_FUSE_BN_SUMS=0 / LHN_GATE_BN_SUMS=0 switch the two off."""
    from litehandnet_amd.plan import BN_BWD, CA_MLP_BWD, DW_BWD, GATE_REDUCE
    _, pb, _ = _build("B", backward=True)
    cb, cf, cbw, nf, nb = pb.finalize()
    bwd = [cbw[i] for i in range(nb)]
    fused = [o for o in bwd if o.kind == DW_BWD and o.ws[4] >= 0]
    skipped = [o for o in bwd if o.kind == BN_BWD and o.i[1] == 1]
    ca = [o for o in bwd if o.kind == CA_MLP_BWD]
    by_ca = {o.ws[k] for o in ca for k in (8, 9) if o.ws[k] >= 0}
    # ... and 19 more get them from ALL their readers' backward kernels (lhn_bnsum: elementwise combines, pools, the fused 1x1
    # backward each add their part): 50 of 52 BatchNorms have no reduce pass (the two left are read by the stride-2 depthwise)
    from litehandnet_amd.plan import AVGPOOL_BWD, EW_BWD, MAXPOOL_BWD, PW_BWD
    by_readers = {o.ws[0] for o in bwd if o.kind in (EW_BWD, MAXPOOL_BWD, PW_BWD) and o.ws[0] >= 0} | \
                 {o.ws[1] for o in bwd if o.kind == AVGPOOL_BWD and o.ws[1] >= 0}
    assert pb.fused_bn_sums == len(fused) == 19 and len(by_ca) == 12 and pb.reader_bn_sums == len(by_readers) == 19 and len(skipped) == 50
    assert {o.ws[4] for o in fused} | by_ca | by_readers == {o.ws[0] for o in skipped}          # the same sums buffers
    for o in fused:
        assert o.i[4] == 1 and o.i[0] == 3 and o.i[1] == 1 and o.i[3] == 1 and o.ws[5] >= 0        # dx stored, 3x3 s1 d1
    for o in ca:
        assert o.ws[5] >= 0 and o.ws[6] >= 0 and o.i[0] > 0           # pooling statistics, saved mean / invstd, packed slice
    assert all(o.ws[4] >= 0 for o in bwd if o.kind == GATE_REDUCE)    # the gate-gradient pass also leaves T0, T1
    fwd = [cf[i] for i in range(nf)]
    assert sum(1 for o in fwd if o.kind == AVGPOOL and o.ws[1] >= 0) == 8
    monkeypatch.setenv("LHN_FUSE_BN_SUMS", "0")
    monkeypatch.setenv("LHN_GATE_BN_SUMS", "0")
    monkeypatch.setenv("LHN_READER_BN_SUMS", "0")
    _, p0, _ = _build("B", backward=True)
    cb, cf, cbw, nf, nb = p0.finalize()
    assert p0.fused_bn_sums == 0 and not any(cbw[i].kind == BN_BWD and cbw[i].i[1] == 1 for i in range(nb))


def test_residual_sum_gradients_ride_in_the_depthwise_backward(monkeypatch):
    """MSRB (litehourglass.py:41-49): `out` feeds the two dilated depthwise halves and the running sum(s).  The sums' output
    gradients are added by those depthwise backward kernels while they store dx (lhn_conv_dw_bwd3): no EW_BWD op writes the
    gradient of such a buffer, every one of its channels is written exactly once (stored), and LHN_GRAD_ADDENDS=0 restores
    the separate passes."""
    from litehandnet_amd.plan import DW_BWD, EW_BWD
    _, pb, _ = _build("B", backward=True)
    cb, cf, cbw, nf, nb = pb.finalize()
    bwd = [cbw[i] for i in range(nb)]
    withadd = [o for o in bwd if o.kind == DW_BWD and (o.ws[2] >= 0 or o.ws[3] >= 0)]
    assert pb.grad_addends == len(withadd) == 8                # 2 MSRBs x 2 rounds x 2 halves
    assert sum(1 for o in withadd if o.ws[3] >= 0) == 4        # the MSRB input collects two sums, the first running sum one
    bufs = {o.in_buf[0] for o in withadd}
    assert not [o for o in bwd if o.kind == EW_BWD and o.in_buf[0] in bufs]
    for b in bufs:
        spans = sorted((o.in_coff[0], o.in_coff[0] + o.in_C[0]) for o in withadd if o.in_buf[0] == b)
        assert spans == [(0, 64), (64, 128)] and all(o.i[4] == 1 for o in withadd if o.in_buf[0] == b)
    grads = {cb[j].grad_off for j in range(len(pb.bufs))}
    assert all(o.ws[2] in grads and (o.ws[3] < 0 or o.ws[3] in grads) for o in withadd)
    def slots(ops):         # gradient destinations of the combines' backward ops (a multi-source op, i[1] == 3, writes 2 or 3)
        return sum((sum(1 for q in range(3) if o.in_buf[q] >= 0) if o.i[1] == 3 else 1) for o in ops if o.kind == EW_BWD)
    n_ew = slots(bwd)
    monkeypatch.setenv("LHN_GRAD_ADDENDS", "0")
    _, p0, _ = _build("B", backward=True)
    _, _, cbw0, _, nb0 = p0.finalize()
    assert p0.grad_addends == 0 and slots([cbw0[i] for i in range(nb0)]) == n_ew + 6


def test_pool_gradients_meet_in_one_store(monkeypatch):
    """The skip tensor of an hourglass level is read by a 2x2 max-pool, an adaptive average pool and the decoder's sum
    (litehourglass.py:139-163): the max-pool's backward op carries the other two gradients (lhn_grad_adds) and their own backward
    ops for that view disappear.  LHN_POOL_GRAD_ADDS=0 restores three read-modify-write passes."""
    from litehandnet_amd.plan import AVGPOOL_BWD, MAXPOOL_BWD
    monkeypatch.delenv("LHN_POOL_GRAD_ADDS", raising=False)
    _, pb, _ = _build("B", backward=True)
    _, _, cbw, _, nb = pb.finalize()
    bwd = [cbw[i] for i in range(nb)]
    fused = [o for o in bwd if o.kind == MAXPOOL_BWD and (o.ws[2] >= 0 or o.ws[3] >= 0)]
    assert pb.pool_grad_adds == len(fused) == 6                       # 2 halves x 3 levels of the hourglass
    assert all(o.ws[2] >= 0 for o in fused)                           # every level has the decoder's sum
    assert sum(1 for o in fused if o.ws[3] >= 0) == 2                 # ... and the top level the 8x8 average pool
    grads = {pb.bufs[j].off["grad"] for j in range(len(pb.bufs)) if "grad" in pb.bufs[j].off}
    assert all(o.ws[2] in grads and (o.ws[3] < 0 or o.ws[3] in grads) for o in fused)
    assert all((o.i[7] >> 16, o.i[7] & 0xffff) == (8, 8) for o in fused if o.ws[3] >= 0)
    n_ap, n_mp = sum(1 for o in bwd if o.kind == AVGPOOL_BWD), sum(1 for o in bwd if o.kind == MAXPOOL_BWD)
    monkeypatch.setenv("LHN_POOL_GRAD_ADDS", "0")
    _, p0, _ = _build("B", backward=True)
    _, _, cbw0, _, nb0 = p0.finalize()
    bwd0 = [cbw0[i] for i in range(nb0)]
    assert p0.pool_grad_adds == 0 and not [o for o in bwd0 if o.kind == MAXPOOL_BWD and (o.ws[2] >= 0 or o.ws[3] >= 0)]
    assert sum(1 for o in bwd0 if o.kind == AVGPOOL_BWD) == n_ap + 2 and sum(1 for o in bwd0 if o.kind == MAXPOOL_BWD) == n_mp
    assert len(bwd0) == len(bwd) + 2 + 6                              # two average-pool and six sum-source passes more


def test_residual_sums_share_one_backward_pass(monkeypatch):
    """Sources of a combine that have its own resolution get their gradient in ONE op (lhn_ew_bwd_multi, i[1] == 3) instead of one
    op per source; upsampled sources keep their own."""
    from litehandnet_amd.plan import EW_BWD
    monkeypatch.delenv("LHN_EW_BWD_MULTI", raising=False)
    for variant, groups in (("A", 21), ("M", 19)):
        _, pb, _ = _build(variant, backward=True)
        _, _, cbw, _, nb = pb.finalize()
        multi = [cbw[i] for i in range(nb) if cbw[i].kind == EW_BWD and cbw[i].i[1] == 3]
        assert pb.ew_bwd_multi == len(multi) == groups
        for o in multi:
            srcs = [q for q in range(3) if o.in_buf[q] >= 0]
            assert len(srcs) >= 2
            ob = pb.bufs[o.out_buf]
            assert all((pb.bufs[o.in_buf[q]].H, pb.bufs[o.in_buf[q]].W, o.in_C[q]) == (ob.H, ob.W, o.out_C) for q in srcs)
        monkeypatch.setenv("LHN_EW_BWD_MULTI", "0")
        _, p0, _ = _build(variant, backward=True)
        _, _, cbw0, _, nb0 = p0.finalize()
        assert not [1 for i in range(nb0) if cbw0[i].kind == EW_BWD and cbw0[i].i[1] == 3]
        assert nb0 == nb + sum(len([q for q in range(3) if o.in_buf[q] >= 0]) - 1 for o in multi)
        monkeypatch.delenv("LHN_EW_BWD_MULTI")

