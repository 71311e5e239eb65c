"""vsr_set_valid_rows_bound: with a caller-supplied upper bound on the non-padding region rows vsr_prepare*() does not read the row count
back (its one host synchronisation): same tokens as the counted path with the exact bound and with a loose one; a bound that is too small
is reported by the input-contract check.  Dense and index-list region formats."""
import numpy as np
import pytest
import torch

from conftest import load_golden
import helpers
from vsrcap import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_bounded_prepare_gives_the_counted_paths_tokens():
    meta, g = load_golden("g2_greedy")
    cfg = dict(meta["cfg"], B=100)
    w = helpers.weights_for(cfg, wseed=meta["wseed"])
    m = helpers.build_model(cfg, w, DEV, bos=meta["bos"])
    det, ctrl = helpers.decode_inputs(cfg, meta["seed"], n=100)
    exact = int((ctrl.sum(-1) != 0).sum())
    total = ctrl.shape[0] * ctrl.shape[1] * ctrl.shape[2]
    det, ctrl = det.to(DEV), ctrl.to(DEV)
    m._engine(torch.device(DEV)).check_ids = True          # (the input-contract check is opt-in: it synchronises)
    with torch.no_grad():
        w0, g0 = m.test(det, ctrl)
        (b0, bg0), _ = m.beam_search((det, ctrl), meta["eos"], 5, 1)
        np.testing.assert_array_equal(w0.cpu().numpy(), g["words"][:100].astype(np.int64))
        for bound in (exact, exact + 777, total, total + 5):
            m.set_valid_rows_bound(bound)
            w1, g1 = m.test(det, ctrl)
            (b1, bg1), _ = m.beam_search((det, ctrl), meta["eos"], 5, 1)
            assert torch.equal(w0, w1) and torch.equal(g0, g1), bound
            assert torch.equal(b0, b1) and torch.equal(bg0, bg1), bound
            m._engine(torch.device(DEV)).raise_on_bad_ids(torch.device(DEV), "bounded prepare")      # nothing to report
        m.set_valid_rows_bound(exact - 10)
        m.test(det, ctrl)
        with pytest.raises(IndexError):
            m._engine(torch.device(DEV)).raise_on_bad_ids(torch.device(DEV), "bound too small")
        m.set_valid_rows_bound(None)
        w2, _ = m.test(det, ctrl)
        assert torch.equal(w0, w2)


def test_bounded_prepare_index_lists():
    from vsrcap.regions import IndexedRegions
    meta, _ = load_golden("g2_greedy")
    cfg = dict(meta["cfg"], B=24)
    w = helpers.weights_for(cfg, wseed=meta["wseed"])
    m = helpers.build_model(cfg, w, DEV, bos=meta["bos"])
    det = torch.from_numpy(synth.make_detections(cfg["B"], cfg["R0"], cfg["D"], seed=5, min_valid=cfg["R0"])).to(DEV)
    idx = torch.from_numpy(synth.make_slot_indices(cfg["B"], cfg["L"], cfg["R"], cfg["R0"], seed=5)).contiguous().to(DEV)
    reg = IndexedRegions(det, idx)
    with torch.no_grad():
        (b0, g0), _ = m.beam_search((det, reg), meta["eos"], 5, 1)
        m.set_valid_rows_bound(cfg["B"] * cfg["R0"])          # every bank row may be non-zero
        (b1, g1), _ = m.beam_search((det, reg), meta["eos"], 5, 1)
    assert torch.equal(b0, b1) and torch.equal(g0, g1)


def _xe_grads(m, det, regions, caps, gts):
    import vsr_oracle as vo
    m.train()
    m.zero_grad()
    out, gate = m((det,), (caps, regions))
    loss = vo.xe_loss(out, gate, caps, gts)[0]
    loss.backward()
    return loss.item(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}


@pytest.mark.parametrize("dtype", ["f16x2", "f32x3", "f32", "bf16"])
def test_training_under_a_row_bound_equals_the_counted_path(dtype):
    """Round-5 advisor finding (high): under a row bound the row list is padded to the bound with copies of its first entry; the forward
    pass stops at the device-side count, the BACKWARD pass gathered all `bound` rows and added (bound - n) extra copies of
    dP[vlist[0]]^T x regions[vlist[0]] to att_va's gradient.  Exact bound, loose bounds (the f16x2 image geometry follows bound % 8) and
    no bound must give the same 28 gradients - bitwise: the padded rows now enter the reduction as zeros."""
    meta, _ = load_golden("g1_xe_wide")
    cfg = meta["cfg"]
    w = helpers.weights_for(cfg, gains=meta["gains"])
    det, ctrl_seq, caps, gts = (x.to(DEV) for x in helpers.train_inputs(cfg, meta["seed"]))
    exact = int((ctrl_seq.sum(-1) != 0).sum())
    total = ctrl_seq.shape[0] * ctrl_seq.shape[1] * ctrl_seq.shape[2]
    assert exact < total - 16
    m = helpers.build_model(cfg, w, DEV).set_compute_dtype(dtype)
    l0, g0 = _xe_grads(m, det, ctrl_seq, caps, gts)
    for bound in (exact, exact + 1, exact + 13, total):
        m.set_valid_rows_bound(bound)
        l1, g1 = _xe_grads(m, det, ctrl_seq, caps, gts)
        assert l1 == l0, (bound, l1, l0)
        for k in g0:
            scale = float(g0[k].abs().max()) + 1e-30
            # the K extent of att_va's weight-gradient GEMM follows the bound (zero columns): other k-pieces, same sum to rounding
            tol = 0.0 if k != "att_va.weight" else 2e-5 * scale
            assert float((g1[k] - g0[k]).abs().max()) <= tol, (bound, k)
    m.set_valid_rows_bound(None)


def test_a_bound_that_is_too_small_zeroes_the_unprojected_rows_and_is_counted():
    """Round-5 advisor finding (medium): rows beyond a too-small bound kept stale workspace floats as their att_va projection.  They are
    zeroed now (a defined result: the run repeats bit for bit whatever the workspace held before) and counted into vsr_bad_ids()."""
    meta, _ = load_golden("g2_greedy")
    cfg = dict(meta["cfg"], B=16)
    w = helpers.weights_for(cfg, wseed=meta["wseed"])
    m = helpers.build_model(cfg, w, DEV, bos=meta["bos"])
    det, ctrl = helpers.decode_inputs(cfg, meta["seed"], n=16)
    exact = int((ctrl.sum(-1) != 0).sum())
    det, ctrl = det.to(DEV), ctrl.to(DEV)
    eng = m._engine(torch.device(DEV))
    with torch.no_grad():
        m.set_valid_rows_bound(exact - 40)
        a = m.test(det, ctrl)
        eng._ws.fill_(0x7f)                       # poison the workspace (NaN-ish floats) and decode again
        m.invalidate_cache()
        b = m.test(det, ctrl)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        eng.check_ids = True
        with pytest.raises(IndexError):
            eng.raise_on_bad_ids(torch.device(DEV), "bound too small")
    m.set_valid_rows_bound(None)


def test_bad_slot_indices_under_a_bound_are_counted():
    """index lists + a row bound: bad slot indices cannot fail the prepare call (nothing is read back); they join vsr_bad_ids()'s count"""
    from vsrcap.regions import IndexedRegions
    meta, _ = load_golden("g2_greedy")
    cfg = dict(meta["cfg"], B=8)
    w = helpers.weights_for(cfg, wseed=meta["wseed"])
    m = helpers.build_model(cfg, w, DEV, bos=meta["bos"])
    det = torch.from_numpy(synth.make_detections(cfg["B"], cfg["R0"], cfg["D"], seed=5, min_valid=cfg["R0"])).to(DEV)
    idx = torch.from_numpy(synth.make_slot_indices(cfg["B"], cfg["L"], cfg["R"], cfg["R0"], seed=5)).contiguous().to(DEV)
    idx[3, 2, 0] = cfg["R0"] + 4                  # outside the feature bank
    reg = IndexedRegions(det, idx)
    with torch.no_grad():
        with pytest.raises(RuntimeError):          # counted path: the prepare call itself fails
            m.test(det, reg)
        m.set_valid_rows_bound(cfg["B"] * cfg["R0"])
        eng = m._engine(torch.device(DEV))
        eng.check_ids = True
        with pytest.raises(IndexError):
            m.test(det, reg)                       # (greedy without verbs does not check by itself ...)
            eng.raise_on_bad_ids(torch.device(DEV), "bad slot index under a bound")
    m.set_valid_rows_bound(None)
