"""Experiment: decode one batch of 100 images as S concurrent sub-batches, one handle + HIP stream + host thread each.

    python tools/dual_stream.py [S ...]

Independent images never exchange data, so sub-batches may overlap freely: the pointwise kernels and the
GEMM prologues/tails of one sub-batch run under the GEMMs of another.  Prints tokens/s for each S.
"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vsr-guided-cic_amd"))
import torch  # noqa: E402
from vsrcap import synth  # noqa: E402
from models import ControllableCaptioningModel  # noqa: E402

CFG = dict(B=100, R0=36, R=36, D=2048, L=10, T=20, V=10000, E=1000, H=1000, A=512)


def main():
    splits = [int(a) for a in sys.argv[1:]] or [1, 2, 4]
    dev = torch.device("cuda", 0)
    c = CFG
    weights = synth.make_weights(c["V"], c["D"], c["E"], c["H"], c["A"], seed=0)
    sd = {k: torch.from_numpy(v) for k, v in weights.items()}
    batches = []
    for i in range(2):
        batches.append((torch.from_numpy(synth.make_detections(c["B"], c["R0"], c["D"], seed=1000 + i)).to(dev),
                        torch.from_numpy(synth.make_ctrl(c["B"], c["L"], c["R"], c["D"], seed=1000 + i)).to(dev)))
    steps, warm = 12, 3
    ref = None
    for S in splits:
        models = []
        for s in range(S):
            m = ControllableCaptioningModel(c["T"], c["V"], 2, det_feat_size=c["D"], input_encoding_size=c["E"],
                                            rnn_size=c["H"], att_size=c["A"], verb_2_vob_all={})
            m.load_state_dict(sd)
            models.append(m.to(dev).eval())
        streams = [torch.cuda.Stream(dev) for _ in range(S)]
        bounds = [(c["B"] * s // S, c["B"] * (s + 1) // S) for s in range(S)]
        parts = [[(d[a:b].contiguous(), r[a:b].contiguous()) for (a, b) in bounds] for d, r in batches]
        outs = [None] * S

        def worker(s, n, first):
            with torch.cuda.stream(streams[s]), torch.no_grad():
                for i in range(first, first + n):
                    det, ctrl = parts[i & 1][s]
                    outs[s] = models[s].beam_search((det, ctrl), [3, -1], 5, 1)[0][0]
                streams[s].synchronize()

        def run(n, first):
            th = [threading.Thread(target=worker, args=(s, n, first)) for s in range(S)]
            for t in th:
                t.start()
            for t in th:
                t.join()

        run(warm, 0)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        run(steps, warm)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        words = torch.cat(outs, 0).cpu()
        if ref is None:
            ref = words
        same = bool((words == ref).all())
        print("S=%d  %.1f tokens/s  %.2f ms/step  tokens identical to S=%d: %s" %
              (S, c["B"] * c["T"] * steps / dt, dt / steps * 1e3, splits[0], same), flush=True)
        del models


if __name__ == "__main__":
    main()
