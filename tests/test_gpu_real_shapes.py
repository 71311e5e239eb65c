"""GPU parity at the shapes the reference's REAL callers feed (the BASELINE configs are synthetic 36 / 36):
  detections (B, 100, 2048)                      /root/reference/data/field.py:115
  slots of 20 regions                            /root/reference/data/field.py:18
  fixed_len 20 (training) / 10 (evaluation)      /root/reference/coco_scripts/train.py:39-41, eval_coco.py:55-57
  ~5 caption rows per image, one beam_search_v call per image with the image's detections expanded over its rows
                                                 /root/reference/coco_scripts/eval_coco.py:240-247
Fixture g12_real_shapes (tests/golden/make_golden.py `real`): the reference's own outputs at these shapes - R0 != R, which no
other full-size fixture exercises.  Runs in every GEMM flavour (tests/conftest.py).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
import helpers
import vsr_oracle as vo
from vsrcap import synth, evalbatch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_xe_step_real_caller_shapes_vs_reference():
    """XE step at B = 100, R0 = 100, R = 20, L = T = 20: loss within 1e-4, the 28 gradient norms within 2e-3 of the reference's autograd."""
    meta, g = load_golden("g12_real_shapes")
    cfg = meta["cfg_xe"]
    assert (cfg["B"], cfg["R0"], cfg["R"], cfg["L"], cfg["T"]) == (100, 100, 20, 20, 20)
    w = helpers.weights_for(cfg, gains=meta["gains_xe"], wseed=meta["wseed"])
    m = helpers.build_model(cfg, w, DEV, bos=meta["bos"])
    det, ctrl_seq, caps, gts = helpers.train_inputs(cfg, meta["seed_xe"])
    m.train()
    m.zero_grad()
    out, gate = m((det.to(DEV),), (caps.to(DEV), ctrl_seq.to(DEV)))
    loss, lc, lg = vo.xe_loss(out, gate, caps.to(DEV), gts.to(DEV))       # train.py:106-110 arithmetic
    loss.backward()
    assert abs(loss.item() - g["xe_losses"][0]) < 1e-4, (loss.item(), g["xe_losses"])
    assert abs(lc.item() - g["xe_losses"][1]) < 1e-4 and abs(lg.item() - g["xe_losses"][2]) < 1e-4
    o, gt_ = out.detach().cpu(), gate.detach().cpu()
    np.testing.assert_allclose(gt_.numpy(), g["xe_gate"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(o[:, :-1].gather(2, caps[:, 1:, None])[:, :, 0].numpy(), g["xe_out_at_target"], atol=2e-5, rtol=0)
    np.testing.assert_array_equal(o.argmax(-1).numpy(), g["xe_out_argmax"])
    grads = {k: p.grad for k, p in m.named_parameters()}
    gn = np.array([float(grads[k].double().norm()) for k in meta["param_order"]])
    np.testing.assert_allclose(gn, g["xe_grad_norm"], rtol=2e-3, atol=1e-8)
    gs = np.array([float(grads[k].double().sum()) for k in meta["param_order"]])
    np.testing.assert_allclose(gs, g["xe_grad_sum"], rtol=5e-3, atol=5e-5)


def _eval_items(meta):
    ce = meta["cfg_eval"]
    n_img, n_caps, seed = meta["n_img"], meta["n_caps"], meta["seed_eval"]
    det = torch.from_numpy(synth.make_detections(n_img, ce["R0"], ce["D"], seed=seed)).to(DEV)
    seqs = torch.from_numpy(synth.make_ctrl(n_img * n_caps, ce["L"], ce["R"], ce["D"], seed=seed)).to(DEV)
    verbs = torch.from_numpy(synth.make_verbs(n_img * n_caps, ce["L"], meta["nv"], seed=seed, p=meta["verb_p"])).to(DEV)
    return det, seqs, verbs


@pytest.mark.parametrize("gt", [False, True])
def test_eval_caller_beam_search_v_batched_vs_reference(gt):
    """16 images x 5 caption rows (R0 = 100, L = 10, R = 20, verbs, beam 5): ONE batched call against the reference's 16 per-image
    beam_search_v calls.  Rows that meet no verb... every row is compared: the verb-forced steps are exact by construction and the
    no-verb twin of these rows agrees between the fp32 reference and the fp64 oracle on every row (eval_noverb_agree64)."""
    meta, g = load_golden("g12_real_shapes")
    ce = meta["cfg_eval"]
    assert (ce["R0"], ce["R"], ce["L"]) == (100, 20, 10) and g["eval_noverb_agree64"].all()
    w = helpers.weights_for(ce, wseed=meta["wseed"])
    m = helpers.build_model(ce, w, DEV, bos=meta["bos"], verb_table=synth.make_verb_table(meta["nv"], ce["V"], seed=0))
    det, seqs, verbs = _eval_items(meta)
    n_caps = meta["n_caps"]
    items = [(det[i], seqs[i * n_caps:(i + 1) * n_caps], verbs[i * n_caps:(i + 1) * n_caps]) for i in range(meta["n_img"])]
    with torch.no_grad():
        res = evalbatch.beam_search_v_batched(m, items, meta["eos"], beam_size=5, out_size=1, gt=gt)
    words = torch.cat([r[0][0] for r in res]).cpu().numpy()
    gates = torch.cat([r[0][1] for r in res]).cpu().numpy()
    np.testing.assert_array_equal(words, g["eval_words_gt%d" % gt].astype(np.int64))
    np.testing.assert_array_equal(gates, g["eval_gates_gt%d" % gt].astype(np.int64))


def test_eval_caller_shapes_without_verbs_vs_reference():
    """the same 80 rows through plain beam_search (every row solid: fp32 reference == fp64 oracle)."""
    meta, g = load_golden("g12_real_shapes")
    ce = meta["cfg_eval"]
    w = helpers.weights_for(ce, wseed=meta["wseed"])
    m = helpers.build_model(ce, w, DEV, bos=meta["bos"])
    det, seqs, _ = _eval_items(meta)
    with torch.no_grad():
        (bw, bg), _ = m.beam_search((det.repeat_interleave(meta["n_caps"], 0).contiguous(), seqs), meta["eos"], 5, 1)
    np.testing.assert_array_equal(bw.cpu().numpy(), g["eval_words_noverb"].astype(np.int64))
    np.testing.assert_array_equal(bg.cpu().numpy(), g["eval_gates_noverb"].astype(np.int64))
