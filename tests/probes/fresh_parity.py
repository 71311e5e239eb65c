"""Full-size parity probe on FRESH seeds (no arg-max margin search, unlike the golden fixtures): beam-5 and greedy tokens of the
HIP path against the CPU oracle for 2 x 48 images (GPU box only; last run: 0 captions differ)."""
import sys, os, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(R, "vsr-guided-cic_amd")); sys.path.insert(0, os.path.join(R, "oracle"))
import torch, numpy as np
from vsrcap import synth
from models import ControllableCaptioningModel
import vsr_oracle as vo
c = dict(B=48, R0=36, R=36, D=2048, L=10, T=20, V=10000, E=1000, H=1000, A=512)
w = synth.make_weights(c["V"], c["D"], c["E"], c["H"], c["A"], seed=0)
m = ControllableCaptioningModel(c["T"], c["V"], 2, det_feat_size=c["D"], input_encoding_size=c["E"], rnn_size=c["H"], att_size=c["A"], verb_2_vob_all={})
m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); m = m.cuda().eval()
o = vo.Oracle(w, c["T"], 2, as_written=False)
torch.set_num_threads(64)
bad = 0
for seed in (901, 902):
    det = torch.from_numpy(synth.make_detections(c["B"], c["R0"], c["D"], seed=seed))
    ctrl = torch.from_numpy(synth.make_ctrl(c["B"], c["L"], c["R"], c["D"], seed=seed))
    with torch.no_grad():
        t0 = time.time()
        (ow, og), _ = o.beam_search(det, ctrl, [3, -1], 5, 1)
        gw, gg = o.test(det, ctrl)
        t1 = time.time()
        (w5, g5), _ = m.beam_search((det.cuda(), ctrl.cuda()), [3, -1], 5, 1)
        wg, ggr = m.test(det.cuda(), ctrl.cuda())
    nb = int((w5.cpu() != ow).any(1).sum()) + int((g5.cpu() != og).any(1).sum())
    ng = int((wg.cpu() != gw).any(1).sum()) + int((ggr.cpu() != gg).any(1).sum())
    bad += nb + ng
    print("seed %d: beam-5 captions differing %d / %d, greedy %d / %d (oracle %.0f s)" % (seed, nb, c["B"], ng, c["B"], t1 - t0), flush=True)
print("FRESH PARITY", "OK" if bad == 0 else "MISMATCH")
