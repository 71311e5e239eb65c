"""How many rows of fixture g3_beam differ from the device's beam-5 captions, and on which kind of row (solid = the fp32 reference and the
fp64 oracle agree)?  python tests/probes/beam_solid_probe.py   (GPU box; environment knobs of the library pass through)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "vsr-guided-cic_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from conftest import load_golden
import helpers

meta, _ = load_golden("g2_greedy")
meta4, _ = load_golden("g4_beam_v")
_, g = load_golden("g3_beam")
cfg = meta["cfg"]
w = helpers.weights_for(cfg, gains=None, wseed=meta.get("wseed", 0))
for dt in sys.argv[1:] or ["f16x2"]:
    m = helpers.build_model(cfg, w, "cuda", bos=meta["bos"], verb_table=meta4["verb_table"]).set_compute_dtype(dt)
    det, ctrl = helpers.decode_inputs(cfg, meta["seed"])
    with torch.no_grad():
        (wd, gate), _ = m.beam_search((det.cuda(), ctrl.cuda()), meta["eos"], 5, 1)
    wd, gate = wd.cpu().numpy(), gate.cpu().numpy()
    same = (wd == g["words"]).all(1) & (gate == g["gates"]).all(1)
    solid = g["agree64"].astype(bool)
    bad = np.nonzero(~same)[0]
    print(dt, "rows differing:", len(bad), "of", len(same), "| on solid rows:", int((~same & solid).sum()), "| rows", bad[:10].tolist())
    for r in bad[:3]:
        print("   row", r, "solid", bool(solid[r]), "first differing step", int(np.nonzero((wd[r] != g["words"][r]) | (gate[r] != g["gates"][r]))[0][0]))
