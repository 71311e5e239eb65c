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
