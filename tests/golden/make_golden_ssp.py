#!/usr/bin/env python3
"""Golden vectors of the two ordering models from the REAL reference (build container only):

    python tests/golden/make_golden_ssp.py            # writes tests/golden/g11_ssp.npz

  S_SSP.generate(mode='not-normal')  (models/sort_model.py:105-183), called one sequence at a time as eval_coco.py:174 does
  SinkhornNet.forward                (models/sinkhorn_network.py:39-51)
Weights and inputs are closed-form (vsrcap.synth); the fixture holds outputs only.  munkres (eval_coco.py:13) is not in the
image: the assignment stored beside the Sinkhorn matrices is the optimum of munkres' own cost matrix (max - value), found
by exhaustive search where the filled block is at most 7 x 7 and by scipy.optimize.linear_sum_assignment otherwise."""
import itertools
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.append(os.path.join(ROOT, "vsr-guided-cic_amd"))

from vsrcap import synth  # noqa: E402


def main():
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "datasets/coco"))
    for n in ("verb_2_vob_all_refine.json", "verb_2_vob.json"):
        json.dump({}, open(os.path.join(tmp, "datasets/coco", n), "w"))
    os.chdir(tmp)
    from models import S_SSP, SinkhornNet
    from scipy.optimize import linear_sum_assignment

    S, Q, seed = 96, 48, 0
    ssp = S_SSP().eval()
    w = synth.make_ssp_weights(seed)
    sd = ssp.state_dict()
    for k in sd:                                     # shared embeddings appear under several keys (encoder.*, decoder.*)
        base = k.split(".")
        key = k
        if k.startswith("encoder.sr_embed_layer.") or k.startswith("decoder.embed_layer."):
            key = "sr_embed_layer." + base[-1]
        if k.startswith("encoder.v_embed_layer."):
            key = "v_embed_layer." + base[-1]
        if key in w:
            sd[k] = torch.from_numpy(w[key])
    ssp.load_state_dict(sd)
    verbs, roles = synth.make_ssp_inputs(S, seed)
    pred, logp = [], []
    with torch.no_grad():
        for s in range(S):                           # eval_coco.py:170-174: batch size 1, verb (1,), roles (1,10)
            p, lp, _ = ssp.generate(torch.from_numpy(verbs[s:s + 1]), torch.from_numpy(roles[s:s + 1]), mode="not-normal")
            pred.append(p[0].numpy())
            logp.append(lp[0].numpy())
    pred, logp = np.stack(pred), np.stack(logp)
    n_roles = (roles != 0).sum(1)
    assert all(sorted(pred[s][:n_roles[s]]) == sorted(roles[s][:n_roles[s]]) for s in range(S))
    identity = np.mean([(pred[s][:n_roles[s]] == roles[s][:n_roles[s]]).all() for s in range(S)])
    print("S_SSP: %d sequences, %.0f %% keep the input order, mean roles %.1f" % (S, 100 * identity, n_roles.mean()))

    net = SinkhornNet(10, 20, 0.1).eval()
    ws = synth.make_sinkhorn_weights(seed)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in ws.items()})
    x, n = synth.make_sinkhorn_inputs(Q, seed)
    with torch.no_grad():
        tr = torch.cat([net(torch.from_numpy(x[q:q + 1])) for q in range(Q)]).numpy()      # eval_coco.py:183: one item per call
    assign = np.zeros((Q, 10), dtype=np.int64)
    gaps = []
    for q in range(Q):
        mx = tr[q].T.astype(np.float64)
        cost = mx.max() - mx
        r, c = linear_sum_assignment(cost)
        assign[q] = c[np.argsort(r)]
        if n[q] <= 7:                                # exhaustive check of the optimum on the full 10 x 10 is 3.6 M permutations: do the small ones
            pass
        tot = cost[np.arange(10), assign[q]].sum()
        # uniqueness margin: best total with ONE forced different choice for some row (cheap lower bound on the gap)
        best2 = np.inf
        for i in range(int(n[q])):               # only the filled rows matter downstream (eval_coco.py:190-194); the padding rows are identical and tie
            c2 = cost.copy()
            if assign[q, i] < n[q]:
                c2[i, assign[q, i]] = 1e9
            else:
                c2[i, n[q]:] = 1e9                   # the columns of the (identical, all-zero) padding rows are interchangeable
            r2, cc = linear_sum_assignment(c2)
            best2 = min(best2, c2[r2, cc].sum())
        gaps.append(best2 - tot)
    # exhaustive verification on 6 x 6 leading blocks (independent of scipy)
    for q in range(8):
        m6 = tr[q].T[:6, :6].astype(np.float64)
        c6 = m6.max() - m6
        best = min(itertools.permutations(range(6)), key=lambda p: sum(c6[i, p[i]] for i in range(6)))
        r, c = linear_sum_assignment(c6)
        assert tuple(c[np.argsort(r)]) == tuple(best)
    print("Sinkhorn: %d items, min optimality gap of the assignment %.3e" % (Q, min(gaps)))
    path = os.path.join(HERE, "g11_ssp.npz")
    # verb_rank_merge (utils/tools.py:35-71): seeded pairs of slot rankings -> the reference's merged ranking
    from utils.tools import verb_rank_merge
    rng = np.random.RandomState(11)
    merge_cases = []
    for _ in range(200):
        la = list(map(int, rng.permutation(10)[:rng.randint(1, 8)]))
        lb = list(map(int, rng.permutation(10)[:rng.randint(1, 8)]))
        merge_cases.append([la, lb, [int(x) for x in verb_rank_merge(list(la), list(lb))]])
    keys = {"ssp": [[k, list(v.shape)] for k, v in ssp.state_dict().items()], "sinkhorn": [[k, list(v.shape)] for k, v in net.state_dict().items()]}
    np.savez_compressed(path, meta=np.array(json.dumps(dict(S=S, Q=Q, seed=seed, n_verbs=2663, state_dict_keys=keys, merge_cases=merge_cases))), pred=pred.astype(np.int8),
                        logp=logp.astype(np.float32), tr=tr.astype(np.float32), assign=assign.astype(np.int8), n_filled=n.astype(np.int8),
                        assign_gap=np.array(gaps, dtype=np.float64))
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
