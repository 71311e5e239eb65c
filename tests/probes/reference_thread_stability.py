"""(build container only: imports /root/reference; not collected by pytest.)  python tests/probes/reference_thread_stability.py [n_seeds]
How stable is the REFERENCE itself under fp32 summation-order changes?  Its beam-5 / greedy ids on fresh captions of g13_flip1024
with 1, 2, 4 and 8 CPU threads (ATen's GEMM blocking and reduction splits change with the thread count), against the fp64 ids."""
import json, os, sys, tempfile, time
import numpy as np, torch
ROOT = "/root/repo"
sys.path.insert(0, "/root/reference"); sys.path.insert(0, ROOT + "/oracle"); sys.path.append(ROOT + "/vsr-guided-cic_amd")
from vsrcap import synth
tmp = tempfile.mkdtemp(); os.makedirs(tmp + "/datasets/coco"); os.chdir(tmp)
json.dump({}, open("datasets/coco/verb_2_vob_all_refine.json", "w")); json.dump({}, open("datasets/coco/verb_2_vob.json", "w"))
from models import ControllableCaptioningModel
z = np.load(ROOT + "/tests/golden/g13_flip1024.npz"); meta = json.loads(str(z["meta"])); c = meta["cfg"]
m = ControllableCaptioningModel(c["T"], c["V"], 2, det_feat_size=c["D"], input_encoding_size=c["E"], rnn_size=c["H"], att_size=c["A"]).eval()
w = synth.make_weights(c["V"], c["D"], c["E"], c["H"], c["A"], seed=0)
m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
nseed = int(sys.argv[1]) if len(sys.argv) > 1 else 8
res = {}
for si, seed in enumerate(meta["seeds"][:nseed]):
    det = torch.from_numpy(synth.make_detections(c["B"], c["R0"], c["D"], seed=seed)); ctrl = torch.from_numpy(synth.make_ctrl(c["B"], c["L"], c["R"], c["D"], seed=seed))
    lo, hi = si * c["B"], (si + 1) * c["B"]
    for nt in (1, 2, 4, 8):
        torch.set_num_threads(nt)
        t0 = time.time()
        with torch.no_grad():
            gw, gg = m.test(det, ctrl)
            (bw, bg), _ = m.beam_search((det, ctrl), [3, -1], 5, 1)
        fg = ((gw.numpy() != z["greedy_words64"][lo:hi]).any(1) | (gg.numpy() != z["greedy_gates64"][lo:hi]).any(1))
        fb = ((bw.numpy() != z["beam_words64"][lo:hi]).any(1) | (bg.numpy() != z["beam_gates64"][lo:hi]).any(1))
        r = res.setdefault(nt, [0, 0, []]); r[0] += int(fg.sum()); r[1] += int(fb.sum()); r[2] += [int(lo + i) for i in np.nonzero(fb)[0]]
        print("seed %d threads %d: greedy flips %d beam flips %d (%.0fs)" % (seed, nt, fg.sum(), fb.sum(), time.time() - t0), flush=True)
print("TOTAL over %d captions:" % (nseed * c["B"]), {k: (v[0], v[1], v[2]) for k, v in res.items()})
