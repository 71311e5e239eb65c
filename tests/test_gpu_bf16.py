"""bf16 THROUGHPUT mode (BASELINE configs[3]; include/vsrcap.h vsr_refresh_bf16_weights): bf16 operands, fp32 accumulation,
fp32 master weights / states / reductions.  It is not a parity mode: these tests state the OBSERVED deviation from the fp32
path (which is pinned to the reference) with explicit, loose tolerances, check that the mode is really active, and that
switching back restores exact parity.

Tolerances (written down, not 1e-4): log-probs of magnitude ~5-60 within 0.15 absolute; XE loss within 1 % relative;
every gradient's cosine similarity with the fp32 gradient >= 0.99 and norm within 3 %; greedy first-token agreement
>= 95 % and whole-sequence token agreement >= 60 % on the 256-sample reference fixture (a flipped near-tie changes the rest
of that caption)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
import helpers
import vsr_oracle as vo

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(meta, gains=None):
    cfg = meta["cfg"]
    w = helpers.weights_for(cfg, gains=gains, wseed=meta.get("wseed", 0))
    return helpers.build_model(cfg, w, DEV, bos=meta["bos"])


def test_bf16_xe_forward_and_gradients_wide():
    meta, g = load_golden("g1_xe_wide")               # B=4, T=12, E=H=1000, A=512, D=512, V=50: the full hidden sizes
    cfg = meta["cfg"]
    m = _model(meta, gains=meta["gains"])
    det, ctrl_seq, caps, gts = helpers.train_inputs(cfg, meta["seed"])
    args = ((det.to(DEV),), (caps.to(DEV), ctrl_seq.to(DEV)))
    res = {}
    for dt in ("f32", "bf16"):
        m.set_compute_dtype(dt)
        m.train()
        m.zero_grad()
        out, gate = m(*args)
        loss = vo.xe_loss(out, gate, caps.to(DEV), gts.to(DEV))[0]
        loss.backward()
        res[dt] = (out.detach().cpu(), gate.detach().cpu(), loss.item(), {k: p.grad.detach().cpu().double().clone() for k, p in m.named_parameters()})
    (o32, g32, l32, gr32), (o16, g16, l16, gr16) = res["f32"], res["bf16"]
    assert abs(l32 - g["losses"][0]) < 1e-4                          # the fp32 pass is still the parity path
    d_out, d_gate = (o16 - o32).abs().max().item(), (g16 - g32).abs().max().item()
    assert 1e-6 < d_out < 0.15 and d_gate < 0.15, (d_out, d_gate)     # different (the mode is active) but close
    assert abs(l16 - l32) < 1e-2 * abs(l32), (l16, l32)
    worst = 1.0
    for k in gr32:
        a, b = gr32[k].flatten(), gr16[k].flatten()
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
        worst = min(worst, cos)
        assert cos >= 0.99, (k, cos)
        assert abs(float(b.norm() / (a.norm() + 1e-30)) - 1.0) < 0.03, k
    print("bf16 vs fp32 (wide config): max |dlogp| words %.3e gates %.3e, loss %.6f vs %.6f, worst gradient cosine %.5f" % (d_out, d_gate, l16, l32, worst))

def test_bf16_xe_step_batch100_full_size_deviation():
    """configs[3] as BASELINE.json states it (XE, batch 100, bf16) at the full sizes - B = 100, T = 20, E = H = 1000, D = 2048,
    V = 10 000: the bf16 throughput mode next to the fp32 parity path on the reference fixture g1_xe_b100.  STATED deviations, bounds =
    1.5 x the values observed on the box (profiles/r04_c_bf16_hoisted_projection_fp32_vs_bf16.txt).  Since round 4 the hoisted
    att_va(regions) projection stays fp32-equivalent in this mode (the shift logit sums up to 36 of its RAW scores, step :187: bf16
    rounding added up coherently there): total loss 19.139088 vs the reference's 19.138969 (1.2e-4: word NLL 9.212266 vs 9.212263, gate
    NLL 2.481705 vs 2.481677; with that GEMM in bf16: 19.1225, 1.6e-2 and a gate-NLL bias of 4e-3), max |d log-prob| 5.4e-4 on words and 3.0e-2 on gates (was 5.6e-2), arg-max of the word
    log-probs equal on 99.15 % of the 2000 rows, every one of the 28 gradients with cosine >= 0.9998 against the fp32 gradient (worst:
    the 512-element att_s.weight) and norm within 2.2 %."""
    meta, g = load_golden("g1_xe_b100")
    cfg = meta["cfg"]
    assert cfg["B"] == 100 and cfg["V"] == 10000 and cfg["H"] == 1000
    m = _model(meta, gains=meta["gains"])
    det, ctrl_seq, caps, gts = helpers.train_inputs(cfg, meta["seed"])
    args = ((det.to(DEV),), (caps.to(DEV), ctrl_seq.to(DEV)))
    res = {}
    for dt in ("f32", "bf16"):
        m.set_compute_dtype(dt)
        m.train()
        m.zero_grad()
        out, gate = m(*args)
        loss, lc, lg = vo.xe_loss(out, gate, caps.to(DEV), gts.to(DEV))
        loss.backward()
        res[dt] = (out.detach().argmax(-1).cpu(), (loss.item(), lc.item(), lg.item()),
                   {k: p.grad.detach().double().flatten().clone() for k, p in m.named_parameters()},
                   out.detach().clone(), gate.detach().clone())
    (a32, l32, gr32, o32, g32), (a16, l16, gr16, o16, g16) = res["f32"], res["bf16"]
    assert abs(l32[0] - g["losses"][0]) < 1e-4                       # the fp32 pass is the parity path
    print("bf16 losses (total, words, gates) %s vs the reference's %s" % (l16, tuple(float(x) for x in g["losses"])))
    assert abs(l16[0] - g["losses"][0]) < 2e-4 and abs(l16[1] - g["losses"][1]) < 2e-5 and abs(l16[2] - g["losses"][2]) < 5e-5, (l16, g["losses"])
    d_out, d_gate = (o16 - o32).abs().max().item(), (g16 - g32).abs().max().item()
    assert d_out > 1e-6, "the bf16 kernels did not run"
    assert d_out < 8.1e-4 and d_gate < 4.5e-2, (d_out, d_gate)
    agree = (a16 == a32).float().mean().item()
    assert agree >= 0.985, agree
    worst, worst_k, worst_n = 1.0, None, 0.0
    for k in gr32:
        a, b = gr32[k], gr16[k]
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-300))
        rn = abs(float(b.norm() / (a.norm() + 1e-300)) - 1.0)
        if cos < worst:
            worst, worst_k = cos, k
        worst_n = max(worst_n, rn)
        assert cos >= 0.9997, (k, cos)
        assert rn < 0.033, (k, rn)
    print("bf16 vs fp32 at B=100 / V=10000: loss %.6f vs %.6f (reference %.6f), max |dlogp| words %.3e gates %.3e, arg-max agreement "
          "%.4f, worst gradient cosine %.6f (%s), worst norm deviation %.4f" % (l16[0], l32[0], g["losses"][0], d_out, d_gate, agree, worst, worst_k, worst_n))


def test_bf16_decode_agreement_and_switch_back_full_size():
    meta, g = load_golden("g2_greedy")
    m = _model(meta)
    det, ctrl = helpers.decode_inputs(meta["cfg"], meta["seed"], n=64)
    det, ctrl = det.to(DEV), ctrl.to(DEV)
    ref_w, ref_g = g["words"][:64].astype(np.int64), g["gates"][:64].astype(np.int64)
    with torch.no_grad():
        m.set_compute_dtype("bf16")
        w16, g16 = m.test(det, ctrl)
        (b16, _), _ = m.beam_search((det, ctrl), meta["eos"], 5, 1)
        m.set_compute_dtype("f32")
        w32, g32 = m.test(det, ctrl)
    np.testing.assert_array_equal(w32.cpu().numpy(), ref_w)          # back in fp32: exact again
    np.testing.assert_array_equal(g32.cpu().numpy(), ref_g)
    w16 = w16.cpu().numpy()
    first = (w16[:, 0] == ref_w[:, 0]).mean()
    every = (w16 == ref_w).mean()
    rows = (w16 == ref_w).all(1).mean()
    print("bf16 greedy vs the reference's fp32 tokens: first token %.3f, all positions %.3f, whole captions %.3f" % (first, every, rows))
    assert first >= 0.95 and every >= 0.60
    assert not np.array_equal(w16, ref_w) or True                    # identical is allowed, divergence is expected
    assert b16.shape == (64, meta["cfg"]["T"]) and int(b16.min()) >= 0 and int(b16.max()) < meta["cfg"]["V"]


def test_bf16_needs_multiples_of_8():
    cfg = dict(V=61, B=3, R0=6, R=5, D=128, L=4, T=7, E=36, H=44, A=12)
    m = helpers.build_model(cfg, helpers.weights_for(cfg), DEV)
    det, ctrl = helpers.decode_inputs(cfg, 3)
    m.set_compute_dtype("bf16")
    with pytest.raises(RuntimeError, match="multiples of 8"):
        m.test(det.to(DEV), ctrl.to(DEV))
    m.set_compute_dtype("f32")
    with torch.no_grad():
        m.test(det.to(DEV), ctrl.to(DEV))


def test_bf16_k_aligned_plan_equals_stream_k_plan(monkeypatch):
    """The 128 x 256 bf16 kernel under its two work decompositions (gemm_plan_aligned: one k-aligned piece of one tile per
    workgroup; gemm_plan: stream-K ranges, VSR_GEMM_ALIGNED=0): the same bf16 products, only the fp32 order in which the k pieces
    of a tile are added differs - log-probs (magnitude 5-60) within 1e-3 of each other (observed 2.3e-4), the same greedy tokens on the first five steps of at least
    90 % of 64 captions (a caption may legitimately follow a flipped near-tie)."""
    meta, _ = load_golden("g1_xe_wide")
    cfg = meta["cfg"]
    det, ctrl_seq, caps, _ = helpers.train_inputs(cfg, meta["seed"])
    args = ((det.to(DEV),), (caps.to(DEV), ctrl_seq.to(DEV)))
    metag, _ = load_golden("g2_greedy")
    detg, ctrlg = helpers.decode_inputs(metag["cfg"], metag["seed"], 64)
    outs, toks = [], []
    for aligned in ("1", "0"):
        monkeypatch.setenv("VSR_GEMM_ALIGNED", aligned)
        m = _model(meta, gains=meta["gains"])            # a fresh model -> a fresh handle that reads the environment
        m.set_compute_dtype("bf16")
        with torch.no_grad():
            out, gate = m(*args)
        outs.append((out.cpu(), gate.cpu()))
        mg = _model(metag)
        mg.set_compute_dtype("bf16")
        with torch.no_grad():
            w, _ = mg.test(detg.to(DEV), ctrlg.to(DEV))
        toks.append(w.cpu())
    assert (outs[0][0] - outs[1][0]).abs().max().item() < 1e-3
    assert (outs[0][1] - outs[1][1]).abs().max().item() < 1e-3
    # (round 4: with the hoisted region projection fp32-equivalent one of the 64 captions follows a flipped near-tie inside its first five
    # steps - bf16 is not a parity mode: the claim is the log-prob bound above, the tokens are a sanity check)
    same = (toks[0][:, :5] == toks[1][:, :5]).all(1).float().mean().item()
    print("greedy captions with identical first five tokens under the two work decompositions: %.3f" % same)
    assert same >= 0.9


def test_bf16_producer_written_a_images_change_nothing(monkeypatch):
    """bf16 mode, decode: the producers of the GEMMs' A operands (k_lstm1, k_attend, k_lstm2) write a bf16 image next to the fp32
    value and the GEMM loads that image instead of converting the fp32 rows itself (VSR_BF16_A16=0: off).  Same rounding of the same
    values: tokens, gates and scores must be IDENTICAL with and without, for greedy and beam search."""
    meta, _ = load_golden("g2_greedy")
    det, ctrl = helpers.decode_inputs(meta["cfg"], meta["seed"], n=48)
    det, ctrl = det.to(DEV), ctrl.to(DEV)
    res = []
    for a16 in ("1", "0"):
        monkeypatch.setenv("VSR_BF16_A16", a16)
        m = _model(meta)                                  # a fresh model -> a fresh handle that reads the environment
        m.set_compute_dtype("bf16")
        with torch.no_grad():
            w, g = m.test(det, ctrl)
            (bw, bg), (lw, lg) = m.beam_search((det, ctrl), meta["eos"], 5, 1)
        res.append((w.cpu(), g.cpu(), bw.cpu(), bg.cpu(), lw.cpu(), lg.cpu()))
    for a, b in zip(*res):
        assert torch.equal(a, b)
