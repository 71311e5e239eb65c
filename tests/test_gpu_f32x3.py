"""'f32x3' GEMM flavour (csrc/gemm_f32x3.h: every fp32 operand split into three bf16 terms, six bf16 MFMAs per product, fp32
accumulation) against the SAME reference fixtures and the SAME bounds as the exact fp32 fma chain: token ids exactly, XE loss
<= 1e-4, gradient norms <= 2e-3, replayed log-probs <= 1e-4 - plus its error against an fp64 oracle next to the chain's.
(The whole GPU suite also runs in this flavour with VSR_COMPUTE_DTYPE=f32x3 python -m pytest tests -m gpu.)"""
import numpy as np
import pytest
import torch

from conftest import load_golden
import helpers
import vsr_oracle as vo
from vsrcap import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(meta, gains=None, dtype="f32x3"):
    cfg = meta["cfg"]
    w = helpers.weights_for(cfg, gains=gains, wseed=meta.get("wseed", 0))
    return helpers.build_model(cfg, w, DEV, bos=meta["bos"]).set_compute_dtype(dtype), w


def test_f32x3_decode_tokens_256_and_fresh_seeds():
    meta, g = load_golden("g2_greedy")
    _, gb = load_golden("g3_beam")
    m, _ = _model(meta)
    det, ctrl = helpers.decode_inputs(meta["cfg"], meta["seed"])
    with torch.no_grad():
        w, gate = m.test(det.to(DEV), ctrl.to(DEV))
        (bw, bg), _ = m.beam_search((det.to(DEV), ctrl.to(DEV)), meta["eos"], 5, 1)
    np.testing.assert_array_equal(w.cpu().numpy(), g["words"].astype(np.int64))
    np.testing.assert_array_equal(gate.cpu().numpy(), g["gates"].astype(np.int64))
    same = (bw.cpu().numpy() == gb["words"]).all(1) & (bg.cpu().numpy() == gb["gates"]).all(1)
    assert same[gb["agree64"].astype(bool)].all() and same.mean() >= 0.99
    metaf, gf = load_golden("g10_fresh")
    for seed in metaf["seeds"]:
        det, ctrl = helpers.decode_inputs(metaf["cfg"], seed)
        with torch.no_grad():
            gw, gg = m.test(det.to(DEV), ctrl.to(DEV))
            (fw, fg), _ = m.beam_search((det.to(DEV), ctrl.to(DEV)), metaf["eos"], 5, 1)
        marg = gf["margins_%d" % seed]
        solid = (marg[:, :, 0].min(1) >= 1e-4) & (marg[:, :, 1].min(1) >= 2e-3)
        ok = (gw.cpu().numpy() == gf["greedy_words_%d" % seed]).all(1) & (gg.cpu().numpy() == gf["greedy_gates_%d" % seed]).all(1)
        assert ok[solid].all()
        okb = (fw.cpu().numpy() == gf["beam_words_%d" % seed]).all(1) & (fg.cpu().numpy() == gf["beam_gates_%d" % seed]).all(1)
        assert okb[gf["beam_agree64_%d" % seed].astype(bool)].all()


def test_f32x3_xe_batch100_loss_and_gradient_norms():
    meta, g = load_golden("g1_xe_b100")
    cfg = meta["cfg"]
    m, _ = _model(meta, gains=meta["gains"])
    det, ctrl_seq, caps, gts = helpers.train_inputs(cfg, meta["seed"])
    m.train()
    m.zero_grad()
    out, gate = m((det.to(DEV),), (caps.to(DEV), ctrl_seq.to(DEV)))
    loss, lc, lg = vo.xe_loss(out, gate, caps.to(DEV), gts.to(DEV))
    loss.backward()
    assert abs(loss.item() - g["losses"][0]) < 1e-4 and abs(lc.item() - g["losses"][1]) < 1e-4 and abs(lg.item() - g["losses"][2]) < 1e-4
    np.testing.assert_allclose(gate.detach().cpu().numpy(), g["gate"], atol=2e-5, rtol=0)
    np.testing.assert_array_equal(out.detach().cpu().argmax(-1).numpy(), g["out_argmax"])
    grads = {k: p.grad for k, p in m.named_parameters()}
    gn = np.array([float(grads[k].double().norm()) for k in meta["param_order"]])
    np.testing.assert_allclose(gn, g["grad_norm"], rtol=2e-3, atol=1e-8)


def test_f32x3_sample_replay_500_rows():
    meta, g = load_golden("g9_scst_500")
    m, _ = _model(meta)
    det, ctrl = helpers.decode_inputs(meta["cfg"], meta["seed"])
    det5 = det.repeat_interleave(meta["n_rep"], 0).contiguous().to(DEV)
    ctrl5 = ctrl.repeat_interleave(meta["n_rep"], 0).contiguous().to(DEV)
    fw, fg = torch.from_numpy(g["words"].astype(np.int64)), torch.from_numpy(g["gates"].astype(np.int64))
    with torch.no_grad():
        _, (lw, lg) = m.sample_rl(det5, ctrl5, forced=(fw, fg))
    np.testing.assert_allclose(lw.cpu().numpy(), g["lp_w"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(lg.cpu().numpy(), g["lp_g"], atol=1e-4, rtol=0)


def test_f32x3_error_against_fp64_next_to_the_fma_chain(monkeypatch):
    """teacher-forced log-probs on the wide fixture (E = H = 1000, A = 512, K up to 2512 per product): |error| against the fp64
    oracle for the exact fp32 chain and for f32x3 - the split must not be the less accurate of the two by more than 1.5x."""
    monkeypatch.setenv("VSR_X3_MIN_ROWS", "1")          # this fixture has 4 rows: by default such launches stay on the exact kernels
    meta, _ = load_golden("g1_xe_wide")
    cfg = meta["cfg"]
    det, ctrl_seq, caps, _ = helpers.train_inputs(cfg, meta["seed"])
    w = helpers.weights_for(cfg, gains=meta["gains"])
    o64 = vo.Oracle(w, cfg["T"], 2, as_written=False, dtype=torch.float64)
    with torch.no_grad():
        ref, refg = o64.forward(det.double(), caps, ctrl_seq.double())
    errs = {}
    for dt in ("f32", "f32x3"):
        m, _ = _model(meta, gains=meta["gains"], dtype=dt)
        with torch.no_grad():
            out, gate = m((det.to(DEV),), (caps.to(DEV), ctrl_seq.to(DEV)))
        errs[dt] = max((out.cpu().double() - ref).abs().max().item(), (gate.cpu().double() - refg).abs().max().item())
    print("max |log-prob error| vs fp64: fma chain %.3e, f32x3 %.3e" % (errs["f32"], errs["f32x3"]))
    assert errs["f32x3"] != errs["f32"], "the f32x3 kernel did not run"
    assert errs["f32x3"] <= 1.5 * errs["f32"] + 1e-6 and errs["f32x3"] < 5e-5
