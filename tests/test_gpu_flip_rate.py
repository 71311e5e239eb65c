"""Flip rate of every fp32 GEMM flavour against the fp64 oracle on 1 024 FRESH captions (no margin search), next to the reference's own.

Fixture g13_flip1024 (tests/golden/make_golden.py `flip1024`): 16 input seeds x 64 images at full size (36 x 2048 regions, 10 slots,
T = 20, V = 10 000); greedy and beam-5 ids of the REFERENCE as it runs in fp32 on the build container's CPU
(/root/reference/models/CaptioningModel.py:38-52, :116-195) and of the fp64 oracle.  A caption "flips" when any of its 20 word ids or 20
gate ids differs from the fp64 ids.  No fp32 implementation can be asked for fewer flips than the reference itself shows, so the bar for
the headline default (f16x2) is: flips <= the reference's count - or it stops being the default.  The other two flavours may flip only rows
the UNMODIFIED reference itself flips at some CPU thread count (the fixture's metadata lists them: `reference_flips_by_cpu_threads`, the
recorded output of tests/probes/reference_thread_stability.py; round 6 - the bound was "+ 2" before) (observed on the box, round 5: reference 0 / 0, f16x2 0 / 0, f32x3 0 / 1, f32 - the exact fp32 fma chain - 0 / 1,
both on row 759 of the beam set: a row on which the fp32 reference and the fp64 oracle agree can still be within one fp32 rounding of a
different beam, which is what "solid" in the 256-caption fixtures cannot promise either; the unmodified reference itself flips row 759
when it runs with ONE CPU thread instead of 2 / 4 / 8: tests/probes/reference_thread_stability.py).
The counts and the flipped rows are printed (pytest -s / the -rA summary) and recorded in DESIGN.md section 2."""
import numpy as np
import pytest
import torch

from conftest import load_golden
import helpers

pytestmark = pytest.mark.gpu
DEV = "cuda"
FLAVOURS = ("f16x2", "f32x3", "f32")
DEFAULT = "f16x2"


def test_flip_rate_per_flavour_on_1024_fresh_captions():
    meta, g = load_golden("g13_flip1024")
    cfg, seeds = meta["cfg"], meta["seeds"]
    n = cfg["B"]
    assert n * len(seeds) == 1024 and g["greedy_words"].shape == (1024, cfg["T"])
    w = helpers.weights_for(cfg, wseed=meta["wseed"])
    ids64 = {"greedy": (g["greedy_words64"].astype(np.int64), g["greedy_gates64"].astype(np.int64)),
             "beam": (g["beam_words64"].astype(np.int64), g["beam_gates64"].astype(np.int64))}
    ref = {"greedy": (g["greedy_words"].astype(np.int64), g["greedy_gates"].astype(np.int64)),
           "beam": (g["beam_words"].astype(np.int64), g["beam_gates"].astype(np.int64))}

    def flips(words, gates, which):
        w64, g64 = ids64[which]
        return np.nonzero((words != w64).any(1) | (gates != g64).any(1))[0]

    ref_flips = {k: flips(*ref[k], k) for k in ref}
    models = {}
    for fl in FLAVOURS:
        m = helpers.build_model(cfg, w, DEV, bos=meta["bos"])
        m.set_compute_dtype(fl)
        models[fl] = m
    out = {fl: {"greedy": ([], []), "beam": ([], [])} for fl in FLAVOURS}
    for seed in seeds:
        det, ctrl = helpers.decode_inputs(cfg, seed)
        det, ctrl = det.to(DEV), ctrl.to(DEV)
        with torch.no_grad():
            for fl, m in models.items():
                gw, gg = m.test(det, ctrl)
                (bw, bg), _ = m.beam_search((det, ctrl), meta["eos"], 5, 1)
                out[fl]["greedy"][0].append(gw.cpu().numpy()); out[fl]["greedy"][1].append(gg.cpu().numpy())
                out[fl]["beam"][0].append(bw.cpu().numpy()); out[fl]["beam"][1].append(bg.cpu().numpy())
    # rows the unmodified reference flips at SOME thread count (1, 2, 4 or 8 CPU threads): an fp32 implementation with another summation
    # order may flip those, and no others
    unstable = {k: set(sum(meta["reference_flips_by_cpu_threads"][k].values(), [])) for k in ("greedy", "beam")}
    report, bad = [], []
    report.append("reference fp32 (CPU): greedy %d / 1024 flips, beam-5 %d / 1024" % (len(ref_flips["greedy"]), len(ref_flips["beam"])))
    for fl in FLAVOURS:
        for which in ("greedy", "beam"):
            words, gates = (np.concatenate(x) for x in out[fl][which])
            f = flips(words, gates, which)
            report.append("%-6s %-6s: %d / 1024 captions differ from the fp64 ids%s" % (fl, which, len(f), (" rows " + str(f.tolist())) if len(f) else ""))
            allowed = set(ref_flips[which].tolist()) | (set() if fl == DEFAULT else unstable[which])
            if (fl == DEFAULT and len(f) > len(ref_flips[which])) or (fl != DEFAULT and not set(f.tolist()) <= allowed):
                bad.append((fl, which, f.tolist()))
    print("\n".join(report))
    assert not bad, "more flips than allowed (default flavour: the fp32 reference's own count; others: only rows the reference itself flips at some CPU thread count): %r\n%s" % (bad, "\n".join(report))
