"""'f16x2' GEMM flavour (csrc/gemm_h2.h: every fp32 operand = two fp16 terms under a power-of-two scale, three fp16 MFMAs per
product, fp32 accumulation, weights pre-split into fp16-pair images) against the SAME reference fixtures and bounds as the exact fma
chain - token ids exactly, XE loss <= 1e-4, gradient norms <= 2e-3, replayed log-probs <= 1e-4 - plus its error against an fp64
oracle next to the chain's and f32x3's, the behaviour of its scale bounds (loosened bounds: fp16 subnormals; huge and tiny operands:
no overflow, no loss), and that the f16x2 kernels are what ran.  (The decoder modules of the suite also run in this flavour:
tests/conftest.py.)"""
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from conftest import load_golden, ROOT
import helpers
import vsr_oracle as vo

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOOL = os.path.join(ROOT, "tools", "gemm_bench")


def _model(meta, gains=None, dtype="f16x2"):
    cfg = meta["cfg"]
    w = helpers.weights_for(cfg, gains=gains, wseed=meta.get("wseed", 0))
    return helpers.build_model(cfg, w, DEV, bos=meta["bos"]).set_compute_dtype(dtype), w


def test_h2_decode_tokens_256_and_fresh_seeds():
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


def test_h2_xe_batch100_loss_and_gradient_norms():
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


@pytest.mark.parametrize("k", [-40, 30])
def test_h2_backward_bounds_follow_the_gradients(k):
    """the backward GEMMs scale their A operands - gradients - by bounds their producers measure on the device: an upstream gradient
    scaled by 2^k moves every bound by k binades, every fp16 pair keeps its bits, and the 28 parameter gradients come out as EXACTLY
    2^k times the unscaled ones (2^-40: a fixed scale would flush the fp16 terms to zero; 2^30: it would overflow them)."""
    meta, _ = load_golden("g1_xe_b100")
    cfg = meta["cfg"]
    m, _ = _model(meta, gains=meta["gains"])
    det, ctrl_seq, caps, gts = helpers.train_inputs(cfg, meta["seed"])
    m.train()

    def grads(scale):
        m.zero_grad()
        out, gate = m((det.to(DEV),), (caps.to(DEV), ctrl_seq.to(DEV)))
        loss, _, _ = vo.xe_loss(out, gate, caps.to(DEV), gts.to(DEV))
        (loss * scale).backward()
        return {n: p.grad.clone() for n, p in m.named_parameters()}
    g1, gk = grads(1.0), grads(2.0 ** k)
    for n in g1:
        assert torch.isfinite(gk[n]).all(), n
        ref = g1[n] * 2.0 ** k
        big = ref.abs() > 2.0 ** -100                  # (fp32 subnormals of the scaled gradient round: not the GEMMs' doing)
        assert torch.equal(gk[n][big], ref[big]), "%s: %d of %d elements differ" % (n, int((gk[n][big] != ref[big]).sum()), int(big.sum()))
        assert float(g1[n].abs().max()) > 0


def test_h2_sample_replay_500_rows():
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


def _wide_errors(dtypes, scale_regions=1.0, scale_weights=1.0):
    meta, _ = load_golden("g1_xe_wide")
    cfg = meta["cfg"]
    det, ctrl_seq, caps, _ = helpers.train_inputs(cfg, meta["seed"])
    det, ctrl_seq = det * scale_regions, ctrl_seq * scale_regions
    w = helpers.weights_for(cfg, gains=meta["gains"])
    if scale_weights != 1.0:       # the region projections see the same products: att_va, the image columns of the input weights and s_fc's partner scale inversely
        w = dict(w)
        for k in ("att_va.weight",):
            w[k] = w[k] * scale_weights
    o64 = vo.Oracle(w, cfg["T"], 2, as_written=False, dtype=torch.float64)
    with torch.no_grad():
        ref, refg = o64.forward(det.double(), caps, ctrl_seq.double())
    errs, outs = {}, {}
    for dt in dtypes:
        m = helpers.build_model(cfg, w, DEV, bos=meta["bos"]).set_compute_dtype(dt)
        with torch.no_grad():
            out, gate = m((det.to(DEV),), (caps.to(DEV), ctrl_seq.to(DEV)))
        assert torch.isfinite(out).all() and torch.isfinite(gate).all(), dt
        errs[dt] = max((out.cpu().double() - ref).abs().max().item(), (gate.cpu().double() - refg).abs().max().item())
        outs[dt] = out.cpu()
    errs["_outs"] = outs
    return errs


def test_h2_error_against_fp64_next_to_the_fma_chain_and_f32x3(monkeypatch):
    """teacher-forced log-probs on the wide fixture (E = H = 1000, A = 512, K up to 2512 per product): |error| against the fp64
    oracle for the exact fp32 chain, f32x3 and f16x2 - the two-term split must not be less accurate than the chain by more than 1.5x."""
    monkeypatch.setenv("VSR_X3_MIN_ROWS", "1")
    errs = _wide_errors(("f32", "f32x3", "f16x2"))
    print("max |log-prob error| vs fp64: fma chain %.3e, f32x3 %.3e, f16x2 %.3e" % (errs["f32"], errs["f32x3"], errs["f16x2"]))
    # (the MAXIMUM can coincide between flavours - it sits on one log-prob whose final fp32 rounding dominates - the outputs cannot)
    o = errs["_outs"]
    n32, n3 = (o["f16x2"] != o["f32"]).sum().item(), (o["f16x2"] != o["f32x3"]).sum().item()
    print("log-probs that differ in their last bits: f16x2 vs fma chain %d, f16x2 vs f32x3 %d of %d" % (n32, n3, o["f32"].numel()))
    assert n32 > 0 and n3 > 0, "the f16x2 kernels did not run"
    assert errs["f16x2"] <= 1.5 * errs["f32"] + 1e-6 and errs["f16x2"] < 5e-5


@pytest.mark.parametrize("scale", [1e-4, 1.0])
def test_h2_bounds_follow_the_operands(scale):
    """region features scaled by 1e-4 (att_va scaled inversely, so the attention scores stay what they were): the measured bounds move
    the power-of-two scales with the operands - the same error level against fp64 as the exact chain, nothing is lost below fp16's range."""
    errs = _wide_errors(("f32", "f16x2"), scale_regions=scale, scale_weights=1.0 / scale)
    print("regions x %g: max |log-prob error| vs fp64: fma chain %.3e, f16x2 %.3e" % (scale, errs["f32"], errs["f16x2"]))
    assert errs["f16x2"] <= 2.0 * errs["f32"] + 2e-6


def test_h2_huge_operands_do_not_overflow():
    """region features x 3e4 (far beyond fp16's 65504 once multiplied by anything): the scales come from the measured bounds, so the
    fp16 terms cannot overflow - finite log-probs (checked in _wide_errors) that agree with the exact chain's."""
    meta, _ = load_golden("g1_xe_wide")
    cfg = meta["cfg"]
    det, ctrl_seq, caps, _ = helpers.train_inputs(cfg, meta["seed"])
    det, ctrl_seq = det * 3e4, ctrl_seq * 3e4
    w = helpers.weights_for(cfg, gains=meta["gains"])
    outs = {}
    for dt in ("f32", "f16x2"):
        m = helpers.build_model(cfg, w, DEV, bos=meta["bos"]).set_compute_dtype(dt)
        with torch.no_grad():
            out, gate = m((det.to(DEV),), (caps.to(DEV), ctrl_seq.to(DEV)))
        assert torch.isfinite(out).all() and torch.isfinite(gate).all(), dt
        outs[dt] = (out.cpu(), gate.cpu())
    agree = (outs["f32"][0].argmax(-1) == outs["f16x2"][0].argmax(-1)).float().mean().item()
    print("regions x 3e4: arg-max agreement with the exact chain %.3f, max |d log-prob| %.3e" % (agree, (outs["f32"][0] - outs["f16x2"][0]).abs().max().item()))
    assert agree >= 0.95


def test_h2_gemm_accuracy_with_loosened_bounds():
    """tools/gemm_bench, K = 1000 product against fp64: rms error of the f16x2 wide kernel with exact bounds and with bounds loosened by
    12 bits (then almost every `lo` term is an fp16 subnormal: the matrix core honours them) next to the exact fp32 MFMA kernel."""
    if not os.path.exists(TOOL):
        pytest.skip("tools/gemm_bench not built")
    rms = {}
    for loose in ("0", "12"):
        r = subprocess.run([TOOL, "64", "256", "4", "5200", "1"], capture_output=True, text=True, timeout=600, env=dict(os.environ, H2_LOOSE=loose))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        acc = re.findall(r"accuracy K=1000 \((.*?)\): max \|err\| (\S+) rms (\S+)", r.stdout)
        assert len(acc) == 2, r.stdout[-2000:]
        rms["chain"] = float(acc[0][2])
        rms[loose] = float(acc[1][2])
    print("rms error vs fp64 at K = 1000: fma chain %.3e, f16x2 %.3e, f16x2 with bounds loosened by 12 bits %.3e" % (rms["chain"], rms["0"], rms["12"]))
    assert rms["0"] <= 1.2 * rms["chain"] and rms["12"] <= 1.2 * rms["chain"]


def test_h2_step_reports_states_outside_the_unit_class():
    """vsr_step in the f16x2 flavour takes the caller's hidden states as unit-class operands (|h| < 2 at the fixed scale 2^15): a state
    beyond that is counted by the input-contract check instead of silently overflowing fp16; a legal state reports nothing."""
    import helpers
    from conftest import load_golden
    from vsrcap import synth
    meta, _ = load_golden("g6_step")
    cfg = meta["cfg"]
    w = helpers.weights_for(cfg, wseed=meta.get("wseed", 0))
    m = helpers.build_model(cfg, w, "cuda", bos=meta["bos"])
    m.set_compute_dtype("f16x2")
    det, ctrl = helpers.decode_inputs(cfg, meta["seed"])
    B, H = cfg["B"], cfg["H"]
    st = [torch.from_numpy((synth.hash_u01(B * H, 50 + i, 9).reshape(B, H) - 0.5).astype(np.float32)).cuda() for i in range(4)]
    k0 = torch.tensor(meta["k0"], device="cuda")
    prev = (torch.tensor(meta["prev_w"], device="cuda"), torch.tensor(meta["prev_g"], device="cuda"))
    eng = m._engine(torch.device("cuda"))
    eng.check_ids = True
    with torch.no_grad():
        m.step(meta["t"], ((st[0], st[1]), (st[2], st[3]), k0), prev, (det.cuda(), ctrl.cuda()), None, mode="feedback")     # fine
        big = st[2].clone()
        big[0, 0] = 3.0
        with pytest.raises(IndexError):
            m.step(meta["t"], ((st[0], st[1]), (big, st[3]), k0), prev, (det.cuda(), ctrl.cuda()), None, mode="feedback")
