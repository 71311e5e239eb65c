"""GPU parity of the ordering models: S-SSP greedy role ordering and Sinkhorn + assignment, batched on the
device behind the reference's class names, against golden outputs of the reference's own modules (g11_ssp.npz) and the oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden
import ssp_oracle as so
from vsrcap import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def canon(assign, n):
    a = np.array(assign[:n], dtype=np.int64)
    a[a >= n] = n
    return a


def _ssp(meta):
    from models import S_SSP
    m = S_SSP()
    w = synth.make_ssp_weights(meta["seed"], meta["n_verbs"])
    sd = m.state_dict()
    alias = {"encoder.sr_embed_layer.weight": "sr_embed_layer.weight", "decoder.embed_layer.weight": "sr_embed_layer.weight",
             "encoder.v_embed_layer.weight": "v_embed_layer.weight"}
    for k in sd:
        kk = alias.get(k, k)
        if kk in w:
            sd[k] = torch.from_numpy(w[kk])
    m.load_state_dict(sd)
    return m.to(DEV).eval(), w


def _sinkhorn(meta):
    from models import SinkhornNet
    m = SinkhornNet(10, 20, 0.1)
    w = synth.make_sinkhorn_weights(meta["seed"])
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    return m.to(DEV).eval(), w


def test_state_dict_keys_are_the_references():
    meta, _ = load_golden("g11_ssp")
    from models import S_SSP, SinkhornNet
    got = [[k, list(v.shape)] for k, v in S_SSP().state_dict().items()]
    assert sorted(map(tuple, map(lambda x: (x[0], tuple(x[1])), got))) == sorted((k, tuple(s)) for k, s in meta["state_dict_keys"]["ssp"])
    got = [[k, list(v.shape)] for k, v in SinkhornNet(10, 20, 0.1).state_dict().items()]
    assert got == meta["state_dict_keys"]["sinkhorn"]


def test_ssp_generate_batched_matches_reference():
    meta, g = load_golden("g11_ssp")
    m, w = _ssp(meta)
    verbs, roles = synth.make_ssp_inputs(meta["S"], meta["seed"], meta["n_verbs"])
    with torch.no_grad():
        pred, logp = m.generate_batch(torch.from_numpy(verbs).to(DEV), torch.from_numpy(roles).to(DEV))
    np.testing.assert_array_equal(pred.cpu().numpy(), g["pred"].astype(np.int64))                 # all 96 role orders, exact
    np.testing.assert_array_equal(np.trunc(logp.cpu().numpy()), g["logp"])                         # the reference's truncated values
    o = so.SSPOracle(w)
    with torch.no_grad():
        _, olp = o.generate(verbs[:24], roles[:24])
    np.testing.assert_allclose(logp[:24].cpu().numpy(), olp.numpy(), atol=1e-4, rtol=0)
    # the reference's call shape (eval_coco.py:174): one sequence, verb (1,), roles (1,10) -> (pred, truncated log-probs, None)
    with torch.no_grad():
        p1, l1, none = m.generate(torch.from_numpy(verbs[5:6]).to(DEV), torch.from_numpy(roles[5:6]).to(DEV), mode='not-normal')
    assert none is None and p1.dtype == torch.int64 and l1.dtype == torch.int64
    np.testing.assert_array_equal(p1.cpu().numpy()[0], g["pred"][5])
    with pytest.raises(RuntimeError):
        m.cpu().generate(torch.from_numpy(verbs[:1]), torch.from_numpy(roles[:1]), mode='not-normal')


def test_sinkhorn_matrix_and_assignment_match_reference():
    meta, g = load_golden("g11_ssp")
    m, _ = _sinkhorn(meta)
    x, n = synth.make_sinkhorn_inputs(meta["Q"], meta["seed"])
    with torch.no_grad():
        tr, a = m.assign(torch.from_numpy(x).to(DEV))
        tr1 = m(torch.from_numpy(x[3:4]).to(DEV))                                                  # eval_coco.py:183 call shape
    np.testing.assert_allclose(tr.cpu().numpy(), g["tr"], atol=5e-6, rtol=2e-4)
    np.testing.assert_allclose(tr1.cpu().numpy()[0], g["tr"][3], atol=5e-6, rtol=2e-4)
    a = a.cpu().numpy()
    for q in range(meta["Q"]):
        assert sorted(a[q]) == list(range(10))                                                     # a permutation
        np.testing.assert_array_equal(canon(a[q], n[q]), canon(g["assign"][q], n[q]))
        mx = g["tr"][q].T.astype(np.float64)
        cost = mx.max() - mx
        assert abs(cost[np.arange(10), a[q]].sum() - cost[np.arange(10), g["assign"][q]].sum()) < 1e-6   # optimal total cost


def test_rank_captions_equals_the_per_caption_reference_flow():
    """vsrcap.evalbatch.rank_captions (one S-SSP call + one Sinkhorn call for the batch) against the reference's per-caption,
    per-verb flow (eval_coco.py:141-221) re-enacted with the CPU oracle networks."""
    from vsrcap.evalbatch import rank_captions, verb_rank_merge
    meta, _ = load_golden("g11_ssp")
    ssp, w = _ssp(meta)
    sh, ws = _sinkhorn(meta)
    o_ssp, o_sh = so.SSPOracle(w, dtype=torch.float64), so.SinkhornOracle(ws, dtype=torch.float64)      # fp64: the decisions and their margins do not depend on the host's BLAS
    N, L, MV = 12, 10, 3
    rng = np.random.RandomState(3)
    control_verb = np.zeros((N, MV), dtype=np.int64)
    det_seqs_v = np.zeros((N, L, MV), dtype=np.int64)
    det_seqs_sr = np.zeros((N, L, MV), dtype=np.int64)
    for n in range(N):
        nv = rng.randint(1, MV + 1)
        control_verb[n, :nv] = rng.choice(np.arange(1, 2600), nv, replace=False)
        for j in range(rng.randint(3, L + 1)):
            vs = rng.permutation(control_verb[n, :nv])[:rng.randint(1, nv + 1)]     # a slot serves each verb at most once
            for k, v in enumerate(vs):
                det_seqs_v[n, j, k] = v
                det_seqs_sr[n, j, k] = rng.randint(1, 7)                # few distinct roles: repeats need the Sinkhorn net
    feats, _ = synth.make_sinkhorn_inputs(N, 7)
    feats[:] = np.abs(synth.hash_u01(feats.size, 77, 7).reshape(feats.shape).astype(np.float32))
    with torch.no_grad():
        got = rank_captions(ssp, sh, control_verb, det_seqs_v, det_seqs_sr, torch.from_numpy(feats).to(DEV))
    want, safe = [], []
    for n in range(N):                                                   # the reference's loop structure, one caption at a time
        verb_ranks, margin = [], float('inf')
        for verb in control_verb[n]:
            if verb == 0:
                break
            roles = np.zeros(L, dtype=np.int64)
            find_sr, sr_find, need = 0, {}, set()
            for j in range(L):
                for k in range(MV):
                    if verb == det_seqs_v[n, j, k] and find_sr < 10:
                        sr = int(det_seqs_sr[n, j, k])
                        if sr not in sr_find:
                            sr_find[sr] = [j]; roles[find_sr] = sr; find_sr += 1
                        else:
                            sr_find[sr].append(j); need.add(sr)
            if find_sr == 0:
                continue
            with torch.no_grad():
                pred, _, mg = o_ssp.generate(np.array([verb]), roles[None], return_margin=True)
            margin = min(margin, float(mg[0]))
            sr_rank = {}
            for sr in need:
                item = np.zeros((1, 10, 2352), dtype=np.float64)
                for j, loc in enumerate(sr_find[sr]):
                    item[0, j] = feats[n, loc]
                with torch.no_grad():
                    tr_item = o_sh.forward(torch.from_numpy(item))
                    a = o_sh.assign(tr_item)[0]
                margin = min(margin, so.assignment_gap(tr_item[0].numpy(), len(sr_find[sr])))
                if sum(int(a[i]) >= len(sr_find[sr]) for i in range(len(sr_find[sr]))) >= 2:
                    margin = 0.0      # two filled rows paired with (identical) padding columns: their order is the solver's tie-break (munkres: unpinned)
                sr_rank[sr] = so.reorder_from_assignment(a, sr_find[sr])
            vr = []
            for sr in pred[0].numpy():
                if sr == 0:
                    break
                vr += list(sr_rank[int(sr)]) if len(sr_find[int(sr)]) != 1 else sr_find[int(sr)]
            verb_ranks.append(vr)
        final = verb_ranks[0] if verb_ranks else []
        for other in verb_ranks[1:]:
            final = verb_rank_merge(final, other)
        want.append([int(v) for v in final])
        safe.append(margin > 1e-4)
    # captions whose every decision (role picks, assignments) is separated by more than 1e-4 in the fp32 CPU oracle must agree
    # exactly; a near-tie may legitimately fall the other way under a different summation order
    # (with random, untrained Sinkhorn weights many items pair two filled rows with padding columns: those captions are skipped)
    assert sum(safe) >= 4, safe
    for n in range(N):
        if safe[n]:
            assert got[n] == want[n], (n, got[n], want[n])
    assert any(len(r) > 3 for r in got)
