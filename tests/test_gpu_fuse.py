"""In-launch combine of the GEMM k-pieces with the pointwise consumer as the last arriver's epilogue (csrc/gemm_epi.h) against the slab
path it replaces (VSR_FUSE=0: every consumer kernel adds the slabs itself).  The combine adds the slabs in the order the consumers add
them, so the two paths must agree BIT FOR BIT - tokens, log-probs, beam scores, teacher-forced log-prob rows - whichever workgroup of a
tile arrives last; any stale read of another workgroup's slab (the hazard of an in-launch hand-over on 8 XCDs with private L2s) shows up
as a difference.  Repeated, at the sizes where the launch plans differ: M = 500 / 100 (stream-K + k-aligned pieces of 128 x 256 and
128 x 128 tiles), 13 images (M = 65), ragged small sizes (epilogues off: sizes not 16-byte clean).
Reference ops replaced: /root/reference/models/controllable_captioning.py:151-154,176-178,181-182."""
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
    for fuse in ("7", "0"):
        old = os.environ.get("VSR_FUSE")
        os.environ["VSR_FUSE"] = fuse
        try:
            m = helpers.build_model(cfg, w, DEV, bos=bos, verb_table=table)
            m.set_compute_dtype("f16x2")
            m._engine(torch.device(DEV))          # the handle reads VSR_FUSE when it is created
        finally:
            if old is None:
                os.environ.pop("VSR_FUSE", None)
            else:
                os.environ["VSR_FUSE"] = old
        ms.append(m)
    return ms


def _same(a, b, what):
    a, b = a.detach().cpu(), b.detach().cpu()
    assert torch.equal(a, b), "%s differs between the combined and the slab path: max |d| %g at %d of %d entries" % (
        what, (a.double() - b.double()).abs().max().item(), int((a != b).sum()), a.numel())


@pytest.mark.parametrize("B", [100, 13])
def test_decode_paths_bit_identical_with_and_without_combine(B):
    meta, _ = load_golden("g2_greedy")
    cfg = dict(meta["cfg"], B=B)
    w = helpers.weights_for(cfg, wseed=meta["wseed"])
    mf, ms = _pair(cfg, w, meta["bos"])
    for rep, seed in enumerate((meta["seed"], 901, 902)):
        det, ctrl = helpers.decode_inputs(cfg, seed)
        det, ctrl = det.to(DEV), ctrl.to(DEV)
        with torch.no_grad():
            for m_rep in range(2):                 # twice per batch: the tickets have to come back to zero
                a, b = mf.test(det, ctrl), ms.test(det, ctrl)
                _same(a[0], b[0], "greedy words"); _same(a[1], b[1], "greedy gates")
                (aw, ag), (alw, alg) = mf.beam_search((det, ctrl), meta["eos"], 5, 2)
                (bw, bg), (blw, blg) = ms.beam_search((det, ctrl), meta["eos"], 5, 2)
                _same(aw, bw, "beam words"); _same(ag, bg, "beam gates"); _same(alw, blw, "beam word log-probs"); _same(alg, blg, "beam gate log-probs")
            caps = torch.from_numpy(synth.make_captions(B, cfg["T"], cfg["V"], seed=seed)).to(DEV)
            seq = torch.from_numpy(synth.make_ctrl(B, cfg["T"], cfg["R"], cfg["D"], seed=seed + 1000)).to(DEV)
            mf.eval(); ms.eval()
            oa, ga = mf((det,), (caps, seq))
            ob, gb = ms((det,), (caps, seq))
            _same(oa, ob, "teacher-forced word log-probs"); _same(ga, gb, "teacher-forced gate log-probs")
            (sw, sg), (lw, lg) = mf.sample_rl(det, ctrl, seed=17 + rep)
            (tw, tg), (mw, mg) = ms.sample_rl(det, ctrl, seed=17 + rep)
            _same(sw, tw, "sampled words"); _same(lw, mw, "sampled word log-probs"); _same(lg, mg, "sampled gate log-probs")


def test_verb_forced_beam_bit_identical_with_and_without_combine():
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


def test_small_ragged_sizes_fall_back_to_the_slab_path():
    """V = 50 (not a multiple of 4) and H = 64: the epilogues need 16-byte-clean problems; the launch must then keep the slab path
    and still match the oracle (smoke-sized)."""
    meta, g = load_golden("g3_beam_small")
    cfg = meta["cfg"]
    w = helpers.weights_for(cfg, wseed=meta.get("wseed", 0))
    mf, ms = _pair(cfg, w, meta["bos"])
    det, ctrl = helpers.decode_inputs(cfg, meta["seed"])
    with torch.no_grad():
        a, b = mf.test(det.to(DEV), ctrl.to(DEV)), ms.test(det.to(DEV), ctrl.to(DEV))
    _same(a[0], b[0], "greedy words (small)")
