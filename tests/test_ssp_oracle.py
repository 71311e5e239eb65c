"""Pin oracle/ssp_oracle.py (S-SSP role ordering + Sinkhorn region ordering) to the outputs of the reference's
own modules (tests/golden/make_golden_ssp.py -> g11_ssp.npz)."""
import json

import numpy as np
import torch

from conftest import load_golden
import ssp_oracle as so
from vsrcap import synth


def canon(assign, n):
    """assigned columns of the filled rows; the columns of the identical all-zero padding rows are interchangeable"""
    a = np.array(assign[:n], dtype=np.int64)
    a[a >= n] = n
    return a


def test_ssp_generate_matches_reference():
    meta, g = load_golden("g11_ssp")
    o = so.SSPOracle(synth.make_ssp_weights(meta["seed"], meta["n_verbs"]))
    verbs, roles = synth.make_ssp_inputs(meta["S"], meta["seed"], meta["n_verbs"])
    with torch.no_grad():
        pred, logp = o.generate(verbs[:40], roles[:40])
    np.testing.assert_array_equal(pred.numpy(), g["pred"][:40])
    # quirk: the reference allocates seqLogprobs with det_seqs_sr.new_zeros (sort_model.py:121): an INTEGER tensor, so the
    # log-probs it returns are truncated toward zero (eval_coco.py ignores them); the oracle keeps the float values
    np.testing.assert_array_equal(np.trunc(logp.numpy()), g["logp"][:40])
    assert (g["pred"] != roles).any()                      # the model really re-orders


def test_sinkhorn_and_assignment_match_reference():
    meta, g = load_golden("g11_ssp")
    o = so.SinkhornOracle(synth.make_sinkhorn_weights(meta["seed"]))
    x, n = synth.make_sinkhorn_inputs(meta["Q"], meta["seed"])
    np.testing.assert_array_equal(n, g["n_filled"])
    with torch.no_grad():
        tr = o.forward(torch.from_numpy(x))
    np.testing.assert_allclose(tr.numpy(), g["tr"], atol=2e-6, rtol=1e-4)
    a = o.assign(tr)
    for q in range(meta["Q"]):
        np.testing.assert_array_equal(canon(a[q], n[q]), canon(g["assign"][q], n[q]))
    assert g["assign_gap"].min() > 1e-4                     # every stored optimum is unique up to the interchangeable columns


def test_drop_in_classes_have_the_references_state_dict():
    meta, _ = load_golden("g11_ssp")
    from models import S_SSP, SinkhornNet
    got = sorted((k, tuple(v.shape)) for k, v in S_SSP().state_dict().items())
    assert got == sorted((k, tuple(s)) for k, s in meta["state_dict_keys"]["ssp"])
    got = [[k, list(v.shape)] for k, v in SinkhornNet(10, 20, 0.1).state_dict().items()]
    assert got == meta["state_dict_keys"]["sinkhorn"]


def test_verb_rank_merge_matches_reference():
    """vsrcap.evalbatch.verb_rank_merge vs utils/tools.py:35-71 run on 200 seeded pairs of rankings"""
    meta, _ = load_golden("g11_ssp")
    from vsrcap.evalbatch import verb_rank_merge
    for la, lb, want in meta["merge_cases"]:
        assert verb_rank_merge(list(la), list(lb)) == want, (la, lb)
