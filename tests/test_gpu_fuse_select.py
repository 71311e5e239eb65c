"""The selection of step t made INSIDE the LSTM1 kernel of step t + 1 (csrc/kernels.h: k_select_lstm1 for beam search, k_select_simple_lstm1
for greedy / sampling / replay; round 5) against the separate launches it replaces (VSR_FUSE_SELECT=0).  Same expressions in the same order:
ids, returned log-probs and beam scores must agree BIT FOR BIT, at M = 500 / 100 (wide and narrow GEMM tiles), 13 images (streaming
kernel), every beam width, verb forcing, and the both-streams-EOS freeze branch.
Reference ops: /root/reference/models/CaptioningModel.py:47,66-70,136-180 (selection), controllable_captioning.py:151-154 (LSTM1 + gates)."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
import helpers
from vsrcap import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _pair(cfg, w, bos, table=None):
    ms = []
    for fuse in ("3", "0"):
        old = os.environ.get("VSR_FUSE_SELECT")
        os.environ["VSR_FUSE_SELECT"] = fuse
        try:
            m = helpers.build_model(cfg, w, DEV, bos=bos, verb_table=table)
            m._engine(torch.device(DEV))          # the handle reads VSR_FUSE_SELECT when it is created
        finally:
            if old is None:
                os.environ.pop("VSR_FUSE_SELECT", None)
            else:
                os.environ["VSR_FUSE_SELECT"] = old
        ms.append(m)
    return ms


def _same(a, b, what):
    a, b = a.detach().cpu(), b.detach().cpu()
    assert torch.equal(a, b), "%s differs between the fused and the separate selection: %d of %d entries" % (what, int((a != b).sum()), a.numel())


@pytest.mark.parametrize("B", [100, 13])
def test_selection_inside_lstm1_is_bit_identical(B):
    meta, _ = load_golden("g2_greedy")
    cfg = dict(meta["cfg"], B=B)
    w = helpers.weights_for(cfg, wseed=meta["wseed"])
    mf, ms = _pair(cfg, w, meta["bos"])
    for rep, seed in enumerate((meta["seed"], 901)):
        det, ctrl = helpers.decode_inputs(cfg, seed)
        det, ctrl = det.to(DEV), ctrl.to(DEV)
        with torch.no_grad():
            a, b = mf.test(det, ctrl), ms.test(det, ctrl)
            _same(a[0], b[0], "greedy words"); _same(a[1], b[1], "greedy gates")
            for beam, out in ((5, 2), (3, 3), (8, 1), (1, 1)):
                (aw, ag), (alw, alg) = mf.beam_search((det, ctrl), meta["eos"], beam, out)
                (bw, bg), (blw, blg) = ms.beam_search((det, ctrl), meta["eos"], beam, out)
                _same(aw, bw, "beam-%d words" % beam); _same(ag, bg, "beam-%d gates" % beam)
                _same(alw, blw, "beam-%d word log-probs" % beam); _same(alg, blg, "beam-%d gate log-probs" % beam)
            # every id is an EOS of its stream from the first step on: the freeze branch of the selection (CaptioningModel.py:143-150)
            (aw, ag), (alw, _) = mf.beam_search((det, ctrl), [int(a[0][0, 0]), int(a[1][0, 0])], 5, 1)
            (bw, bg), (blw, _) = ms.beam_search((det, ctrl), [int(a[0][0, 0]), int(a[1][0, 0])], 5, 1)
            _same(aw, bw, "beam words with early EOS"); _same(alw, blw, "beam log-probs with early EOS")
            (sw, sg), (lw, lg) = mf.sample_rl(det, ctrl, seed=17 + rep)
            (tw, tg), (mw, mg) = ms.sample_rl(det, ctrl, seed=17 + rep)
            _same(sw, tw, "sampled words"); _same(sg, tg, "sampled gates"); _same(lw, mw, "sampled word log-probs"); _same(lg, mg, "sampled gate log-probs")
            (rw, rg), (rlw, rlg) = mf.sample_rl(det, ctrl, forced=(sw, sg))
            (qw, qg), (qlw, qlg) = ms.sample_rl(det, ctrl, forced=(sw, sg))
            _same(rlw, qlw, "replayed word log-probs"); _same(rlg, qlg, "replayed gate log-probs")


def test_verb_forced_selection_inside_lstm1_is_bit_identical():
    meta, _ = load_golden("g4_beam_v")
    cfg = dict(meta["cfg"], B=32)
    w = helpers.weights_for(cfg, wseed=meta["wseed"])
    mf, ms = _pair(cfg, w, meta["bos"], table=meta["verb_table"])
    det, ctrl = helpers.decode_inputs(cfg, meta["seed"], n=32)
    verbs = torch.from_numpy(synth.make_verbs(32, cfg["L"], meta["nv"], seed=meta["seed"], p=meta["verb_p"])).to(DEV)
    det, ctrl = det.to(DEV), ctrl.to(DEV)
    with torch.no_grad():
        for gt in (False, True):
            (aw, ag), _ = mf.beam_search_v((det, ctrl, verbs), meta["eos"], 5, 1, gt=gt)
            (bw, bg), _ = ms.beam_search_v((det, ctrl, verbs), meta["eos"], 5, 1, gt=gt)
            _same(aw, bw, "verb-forced beam words gt=%s" % gt); _same(ag, bg, "verb-forced beam gates")


@pytest.mark.parametrize("H", [72, 136])
def test_selection_inside_lstm1_small_odd_sizes_vs_oracle(H):
    """hidden sizes that are not multiples of the fused kernel's 64-unit slices (72 = one full + one 8-unit slice, 136 = 2 + 8), B = 3,
    every beam width: fused == separate bit for bit, and the ids are the CPU oracle's."""
    import vsr_oracle as vo
    cfg = dict(V=50, B=3, R0=7, R=5, D=64, L=4, T=9, E=64, H=H, A=32)
    w = helpers.weights_for(cfg)
    mf, ms = _pair(cfg, w, 2)
    det, ctrl = helpers.decode_inputs(cfg, 77)
    o = vo.Oracle(w, cfg["T"], 2, as_written=False)
    with torch.no_grad():
        a, b = mf.test(det.to(DEV), ctrl.to(DEV)), ms.test(det.to(DEV), ctrl.to(DEV))
        ow, og = o.test(det, ctrl)
        _same(a[0], b[0], "greedy words"); _same(a[1], b[1], "greedy gates")
        assert torch.equal(a[0].cpu(), ow) and torch.equal(a[1].cpu(), og)
        for beam in (2, 4, 6, 7):
            (aw, ag), (alw, alg) = mf.beam_search((det.to(DEV), ctrl.to(DEV)), [3, -1], beam, 2)
            (bw, bg), (blw, blg) = ms.beam_search((det.to(DEV), ctrl.to(DEV)), [3, -1], beam, 2)
            _same(aw, bw, "beam-%d words" % beam); _same(ag, bg, "beam-%d gates" % beam); _same(alw, blw, "beam-%d log-probs" % beam)
            (obw, obg), _ = o.beam_search(det, ctrl, [3, -1], beam, 2)
            assert torch.equal(aw.cpu()[:, 0], obw[:, 0]) and torch.equal(ag.cpu()[:, 0], obg[:, 0]), beam      # (the top hypothesis is well defined)
