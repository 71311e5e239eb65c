"""Pin the CPU oracle to the golden vectors captured from the real reference (tests/golden/make_golden.py),
and - when /root/reference is mounted (build container only) - to the reference run live."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
import helpers
import vsr_oracle as vo
from vsrcap import synth


def _oracle(meta, gains=None, table=None, as_written=False, dtype=torch.float32):
    cfg = meta["cfg"]
    w = helpers.weights_for(cfg, gains=gains, wseed=meta.get("wseed", 0))
    return vo.Oracle(w, cfg["T"], meta["bos"], verb_table=table, as_written=as_written, dtype=dtype), w


@pytest.mark.parametrize("name", ["g1_xe_small", "g1_xe_wide", "g1_xe_hot_small"])
@pytest.mark.parametrize("as_written", [True, False])
def test_xe_forward_and_loss(name, as_written):
    meta, g = load_golden(name)
    cfg = meta["cfg"]
    o, _ = _oracle(meta, gains=meta["gains"], as_written=as_written)
    det, ctrl_seq, caps, gts = helpers.train_inputs(cfg, meta["seed"])
    with torch.no_grad():
        out, gate = o.forward(det, caps, ctrl_seq)
    tol = 1e-4 if "hot" in name else 1e-5
    np.testing.assert_allclose(gate.numpy(), g["gate"], atol=tol, rtol=0)
    np.testing.assert_allclose(out.numpy(), g["out"], atol=tol, rtol=0)
    loss, lc, lg = vo.xe_loss(out, gate, caps, gts)
    np.testing.assert_allclose([loss.item(), lc.item(), lg.item()], g["losses"], atol=tol * 10, rtol=0)


def test_xe_gradients_match_reference():
    """autograd through the oracle reproduces the reference's per-parameter gradient norms (G1)."""
    meta, g = load_golden("g1_xe_small")
    cfg = meta["cfg"]
    o, w = _oracle(meta, gains=meta["gains"])
    for k in o.p:
        o.p[k].requires_grad_(True)
    det, ctrl_seq, caps, gts = helpers.train_inputs(cfg, meta["seed"])
    out, gate = o.forward(det, caps, ctrl_seq)
    vo.xe_loss(out, gate, caps, gts)[0].backward()
    gn = np.array([float(o.p[k].grad.double().norm()) for k in meta["param_order"]])
    np.testing.assert_allclose(gn, g["grad_norm"], rtol=2e-4, atol=1e-7)


@pytest.mark.parametrize("as_written", [True, False])
def test_small_decode_loops(as_written):
    meta, g = load_golden("g3_beam_small")
    _, gv = load_golden("g4_beam_v_small")
    cfg = meta["cfg"]
    o, _ = _oracle(meta, table=meta["verb_table"], as_written=as_written)
    det, ctrl = helpers.decode_inputs(cfg, meta["seed"])
    with torch.no_grad():
        w, gt_ = o.test(det, ctrl)
        np.testing.assert_array_equal(w.numpy(), g["greedy_words"])
        np.testing.assert_array_equal(gt_.numpy(), g["greedy_gates"])
        (bw, bg), (lw, lg) = o.beam_search(det, ctrl, meta["eos"], 3, 2)
        np.testing.assert_array_equal(bw.numpy(), g["words_b3o2"])
        np.testing.assert_array_equal(bg.numpy(), g["gates_b3o2"])
        np.testing.assert_allclose(lw.numpy(), g["lpw_b3o2"], atol=1e-4, rtol=0)
        np.testing.assert_allclose(lg.numpy(), g["lpg_b3o2"], atol=1e-4, rtol=0)
        (bw, bg), _ = o.beam_search(det, ctrl, meta["eos"], 5, 1)
        np.testing.assert_array_equal(bw.numpy(), g["words_b5"])
        (b1, _), _ = o.beam_search(det, ctrl, meta["eos"], 1, 1)
        np.testing.assert_array_equal(b1.numpy(), g["greedy_words"])
        verbs = torch.from_numpy(synth.make_verbs(cfg["B"], cfg["L"], meta["nv"], seed=meta["seed"], p=meta["verb_p"]))
        for flag in (False, True):
            (vw, vg), _ = o.beam_search(det, ctrl, meta["eos"], 5, 1, verbs=verbs, gt=flag)
            np.testing.assert_array_equal(vw.numpy(), gv["words_gt%d" % flag])
            np.testing.assert_array_equal(vg.numpy(), gv["gates_gt%d" % flag])


def test_single_step_vector():
    meta, g = load_golden("g6_step")
    cfg = meta["cfg"]
    o, _ = _oracle(meta)
    det, ctrl = helpers.decode_inputs(cfg, meta["seed"])
    B, H = cfg["B"], cfg["H"]
    st = [torch.from_numpy((synth.hash_u01(B * H, 50 + i, 9).reshape(B, H) - 0.5).astype(np.float32)) for i in range(4)]
    state = st + [torch.tensor(meta["k0"])]
    prev = (torch.tensor(meta["prev_w"]), torch.tensor(meta["prev_g"]))
    with torch.no_grad():
        (lw, lg), s = o.step(meta["t"], state, prev, det, ctrl)
    np.testing.assert_array_equal(s[4].numpy(), g["k"])
    np.testing.assert_allclose(lw.numpy(), g["logp_w"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(lg.numpy(), g["logp_g"], atol=1e-5, rtol=0)
    for got, key in zip(s[:4], ("h1", "c1", "h2", "c2")):
        np.testing.assert_allclose(got.numpy(), g[key], atol=1e-6, rtol=0)


def test_full_size_greedy_beam_sample_subset():
    """first 24 of the 256 full-size samples (the whole set is checked on the GPU box and at generation time)."""
    meta, g = load_golden("g2_greedy")
    _, gb = load_golden("g3_beam")
    meta5, g5 = load_golden("g5_sample")
    o, _ = _oracle(meta)
    n = 24
    det, ctrl = helpers.decode_inputs(meta["cfg"], meta["seed"], n=n)
    with torch.no_grad():
        w, gate, marg, ks, _ = o.test(det, ctrl, return_trace=True)
        np.testing.assert_array_equal(w.numpy(), g["words"][:n])
        np.testing.assert_array_equal(gate.numpy(), g["gates"][:n])
        np.testing.assert_array_equal(ks.numpy(), g["slots"][:n])
        (bw, bg), _ = o.beam_search(det, ctrl, meta["eos"], 5, 1)
        solid = gb["agree64"][:n].astype(bool)
        assert ((bw.numpy() == gb["words"][:n]).all(1) & (bg.numpy() == gb["gates"][:n]).all(1))[solid].all()
        fw, fg = torch.from_numpy(g5["words"][:n].astype(np.int64)), torch.from_numpy(g5["gates"][:n].astype(np.int64))
        _, (lw, lg) = o.sample_rl(det, ctrl, forced=(fw, fg))
        np.testing.assert_allclose(lw.numpy(), g5["lp_w"][:n], atol=1e-4, rtol=0)
        np.testing.assert_allclose(lg.numpy(), g5["lp_g"][:n], atol=1e-4, rtol=0)
    # the fixture really is diverse and well separated
    assert len(np.unique(g["words"])) >= 200 and 0.2 <= g["gates"].mean() <= 0.8
    assert (g["slots"][:, -1] == meta["cfg"]["L"] - 1).any()
    assert g["margins"][:, :, 0].min() >= 1e-4 and g["margins"][:, :, 1].min() >= 2e-3


def test_step_v_vector():
    """G6-v: the verb-forced step (controllable_captioning.py:192-297), gt False / True, vs the reference's outputs."""
    meta, g = load_golden("g6_step_v")
    cfg = meta["cfg"]
    o, _ = _oracle(meta, table=meta["verb_table"])
    det, ctrl = helpers.decode_inputs(cfg, meta["seed"])
    B, H = cfg["B"], cfg["H"]
    st = [torch.from_numpy((synth.hash_u01(B * H, 50 + i, 9).reshape(B, H) - 0.5).astype(np.float32)) for i in range(4)]
    prev = (torch.tensor(meta["prev_w"]), torch.tensor(meta["prev_g"]))
    verbs = torch.tensor(meta["verbs"], dtype=torch.float64)
    for flag in (False, True):
        with torch.no_grad():
            (lw, lg), s = o.step(meta["t"], st + [torch.tensor(meta["k0"])], prev, det, ctrl, verbs=verbs, gt=flag)
        np.testing.assert_array_equal(s[4].numpy(), g["k_gt%d" % flag])
        np.testing.assert_allclose(lw.numpy(), g["logp_w_gt%d" % flag], atol=1e-5, rtol=0)
        np.testing.assert_allclose(lg.numpy(), g["logp_g_gt%d" % flag], atol=1e-5, rtol=0)
        np.testing.assert_allclose(s[2].numpy(), g["h2_gt%d" % flag], atol=1e-6, rtol=0)
    assert ((g["logp_w_gt0"] == 0).sum(1) == 1).sum() == 3


def test_scst_500_fixture_slice_and_fresh_seed_subset():
    """configs[4] fixture: replaying the reference's draws through the oracle reproduces its log-probs (first 10 of the
    500 rows; all 500 are replayed on the GPU box).  g10_fresh: first 6 images of a fresh seed, greedy + beam-5 tokens."""
    meta, g = load_golden("g9_scst_500")
    cfg = meta["cfg"]
    o, _ = _oracle(meta)
    det, ctrl = helpers.decode_inputs(cfg, meta["seed"], n=2)
    det5, ctrl5 = det.repeat_interleave(meta["n_rep"], 0), ctrl.repeat_interleave(meta["n_rep"], 0)
    n = det5.size(0)
    fw, fg = torch.from_numpy(g["words"][:n].astype(np.int64)), torch.from_numpy(g["gates"][:n].astype(np.int64))
    with torch.no_grad():
        _, (lw, lg) = o.sample_rl(det5, ctrl5, forced=(fw, fg))
    np.testing.assert_allclose(lw.numpy(), g["lp_w"][:n], atol=1e-4, rtol=0)
    np.testing.assert_allclose(lg.numpy(), g["lp_g"][:n], atol=1e-4, rtol=0)
    assert len(np.unique(g["words"])) >= 200                       # the draws are diverse
    meta, g = load_golden("g10_fresh")
    seed = meta["seeds"][0]
    o, _ = _oracle(meta)
    det, ctrl = helpers.decode_inputs(meta["cfg"], seed, n=6)
    with torch.no_grad():
        w, gate = o.test(det, ctrl)
        (bw, bg), _ = o.beam_search(det, ctrl, meta["eos"], 5, 1)
    np.testing.assert_array_equal(w.numpy(), g["greedy_words_%d" % seed][:6])
    np.testing.assert_array_equal(gate.numpy(), g["greedy_gates_%d" % seed][:6])
    np.testing.assert_array_equal(bw.numpy(), g["beam_words_%d" % seed][:6])
    np.testing.assert_array_equal(bg.numpy(), g["beam_gates_%d" % seed][:6])


def test_round5_fixtures_slices():
    """g13_flip1024 (1 024 fresh captions: reference fp32 AND fp64-oracle ids) and g12_real_shapes (R0 = 100, R = 20; 16 images x 5
    caption rows): the oracle reproduces the stored ids on small slices (the whole sets are decoded on the GPU box)."""
    meta, g = load_golden("g13_flip1024")
    cfg, seed = meta["cfg"], meta["seeds"][3]
    lo = 3 * cfg["B"]
    o, w = _oracle(meta)
    det, ctrl = helpers.decode_inputs(cfg, seed, n=4)
    with torch.no_grad():
        gw, gg = o.test(det, ctrl)
        (bw, bg), _ = o.beam_search(det, ctrl, meta["eos"], 5, 1)
    np.testing.assert_array_equal(gw.numpy(), g["greedy_words"][lo:lo + 4])
    np.testing.assert_array_equal(gg.numpy(), g["greedy_gates"][lo:lo + 4])
    np.testing.assert_array_equal(bw.numpy(), g["beam_words"][lo:lo + 4])
    np.testing.assert_array_equal(bg.numpy(), g["beam_gates"][lo:lo + 4])
    # the reference's own count of captions that differ from the fp64 ids on this set: 0 and 0 (the bar the default GEMM flavour is held to)
    assert (g["greedy_words"] == g["greedy_words64"]).all() and (g["beam_words"] == g["beam_words64"]).all()
    assert (g["greedy_gates"] == g["greedy_gates64"]).all() and (g["beam_gates"] == g["beam_gates64"]).all()
    assert len(np.unique(g["greedy_words"])) >= 1000 and 0.2 <= g["greedy_gates"].mean() <= 0.8

    meta, g = load_golden("g12_real_shapes")
    ce, n_caps = meta["cfg_eval"], meta["n_caps"]
    o = vo.Oracle(w, ce["T"], meta["bos"], as_written=False)          # (same closed-form weights: default gains, seed 0)
    det = torch.from_numpy(synth.make_detections(meta["n_img"], ce["R0"], ce["D"], seed=meta["seed_eval"]))
    seqs = torch.from_numpy(synth.make_ctrl(meta["n_img"] * n_caps, ce["L"], ce["R"], ce["D"], seed=meta["seed_eval"]))
    with torch.no_grad():
        (bw, bg), _ = o.beam_search(det[:2].repeat_interleave(n_caps, 0), seqs[:2 * n_caps], meta["eos"], 5, 1)
    np.testing.assert_array_equal(bw.numpy(), g["eval_words_noverb"][:2 * n_caps])
    np.testing.assert_array_equal(bg.numpy(), g["eval_gates_noverb"][:2 * n_caps])
    assert g["eval_noverb_agree64"].all() and g["xe_losses"][0] > 0


@pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="reference not mounted (GPU box)")
def test_live_against_reference(tmp_path, monkeypatch):
    import json
    import subprocess
    import sys
    # run in a subprocess: the reference's package is also called `models`
    code = r'''
import sys, json, os
sys.path.insert(0, "/root/reference"); sys.path.append(%r); sys.path.insert(0, %r)   # the reference's `models` must win
import torch, numpy as np
from vsrcap import synth
import vsr_oracle as vo
from models import ControllableCaptioningModel
c = dict(V=61, B=3, R0=6, R=5, D=128, L=4, T=7, E=32, H=48, A=16)
w = synth.make_weights(c["V"], c["D"], c["E"], c["H"], c["A"], seed=4)
m = ControllableCaptioningModel(c["T"], c["V"], 2, det_feat_size=c["D"], input_encoding_size=c["E"], rnn_size=c["H"], att_size=c["A"]).eval()
m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
det = torch.from_numpy(synth.make_detections(c["B"], c["R0"], c["D"], seed=2)); ctrl = torch.from_numpy(synth.make_ctrl(c["B"], c["L"], c["R"], c["D"], seed=2))
for aw in (True, False):
    o = vo.Oracle(w, c["T"], 2, as_written=aw)
    with torch.no_grad():
        a = m.test(det, ctrl); b = o.test(det, ctrl)
        assert (a[0] == b[0]).all() and (a[1] == b[1]).all()
        (a, _) = m.beam_search((det, ctrl), [3, -1], 4, 2); (b, _) = o.beam_search(det, ctrl, [3, -1], 4, 2)
        assert (a[0] == b[0]).all() and (a[1] == b[1]).all()
print("LIVE-OK")
''' % (os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vsr-guided-cic_amd"),
       os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    (tmp_path / "datasets" / "coco").mkdir(parents=True)
    (tmp_path / "datasets" / "coco" / "verb_2_vob_all_refine.json").write_text("{}")
    (tmp_path / "datasets" / "coco" / "verb_2_vob.json").write_text("{}")
    r = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert "LIVE-OK" in r.stdout, r.stderr[-2000:]


def test_oracle_training_trajectory_small():
    """g14_xe_traj_small: 5 Adam steps of the reference's training loop (coco_scripts/train.py:92-120) - the oracle under the same loop"""
    import helpers
    meta, g = load_golden("g14_xe_traj_small")
    cfg, fs = meta["cfg"], meta["feat_scale"]
    w = helpers.weights_for(cfg, gains=meta["gains"])
    o = vo.Oracle(w, cfg["T"], meta["bos"], as_written=True)
    params = [o.p[k].requires_grad_(True) for k in o.p]
    opt = torch.optim.Adam(params, lr=meta["lr"])
    losses = []
    for i in range(meta["steps"]):
        det, ctrl_seq, caps, gts = helpers.train_inputs(cfg, meta["seed"] + i)
        out, gate = o.forward(det * fs, caps, ctrl_seq * fs)
        loss, lc, lg = vo.xe_loss(out, gate, caps, gts)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append([loss.item(), lc.item(), lg.item()])
    np.testing.assert_allclose(np.array(losses), g["losses"], atol=2e-5, rtol=0)
