"""Index-list regions on the TRAINING path (second half: `COCOControlSequenceField._fill`, data/field.py:44-61, consumed
at coco_scripts/train.py:100-103): the XE / SCST step on `IndexedRegions` (slot entries name rows of the sample's own detection
matrix) against the same step on the dense (B, T, R, D) tensor the index lists stand for.

Forward values must agree to fp32 summation-order noise (att_va runs once per bank row instead of once per copy: the same dot
products in other tiles), all 28 gradients to 2e-5 of their scale - att_va's gradient is a segmented sum over the entries that name a
bank row followed by a GEMM over bank rows instead of a GEMM over entries - and the small config is also checked against the CPU
oracle differentiated by torch autograd on the dense tensor."""
import numpy as np
import pytest
import torch

import helpers
import vsr_oracle as vo
from vsrcap import synth
from vsrcap.regions import IndexedRegions

pytestmark = pytest.mark.gpu
DEV = "cuda"

SMALL = dict(V=60, B=5, R0=9, R=7, D=512, L=8, T=8, E=64, H=64, A=32)
FULL = dict(V=10000, B=100, R0=36, R=36, D=2048, L=20, T=20, E=1000, H=1000, A=512)


def _inputs(cfg, seed, gains=None):
    w = helpers.weights_for(cfg, gains=gains)
    det = torch.from_numpy(synth.make_detections(cfg["B"], cfg["R0"], cfg["D"], seed=seed, min_valid=max(2, cfg["R0"] // 2)))
    idx = torch.from_numpy(synth.make_slot_indices(cfg["B"], cfg["T"], cfg["R"], cfg["R0"], seed=seed))
    caps = torch.from_numpy(synth.make_captions(cfg["B"], cfg["T"], cfg["V"], seed=seed))
    gts = torch.from_numpy(synth.make_gate_gts(cfg["B"], cfg["T"], seed=seed))
    return w, det, idx, caps, gts


def _xe(m, det, regions, caps, gts):
    m.train()
    m.zero_grad()
    out, gate = m((det,), (caps, regions))
    loss = vo.xe_loss(out, gate, caps, gts)[0]
    loss.backward()
    return out.detach(), gate.detach(), loss.item(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}


def _close(a, b, rel):
    scale = max(float(b.abs().max()), 1e-12)
    return float((a - b).abs().max()) <= rel * scale


@pytest.mark.parametrize("cfg", [SMALL, FULL], ids=["small", "batch100_full_size"])
def test_xe_step_on_index_lists_equals_dense(cfg):
    gains = {k: 1.0 for k in synth.DEFAULT_GAINS}
    w, det, idx, caps, gts = _inputs(cfg, 21, gains)
    m = helpers.build_model(cfg, w, DEV)
    det, idx, caps, gts = det.to(DEV), idx.to(DEV), caps.to(DEV), gts.to(DEV)
    reg = IndexedRegions(det, idx)                     # the bank of a training sample IS its detection matrix
    dense = reg.dense().contiguous()
    assert dense.shape == (cfg["B"], cfg["T"], cfg["R"], cfg["D"])
    o_d, g_d, l_d, gr_d = _xe(m, det, dense, caps, gts)
    o_i, g_i, l_i, gr_i = _xe(m, det, reg, caps, gts)
    assert abs(l_i - l_d) < 2e-5 * max(1.0, abs(l_d)), (l_i, l_d)
    assert float((o_i - o_d).abs().max()) < 2e-4 and float((g_i - g_d).abs().max()) < 2e-4
    for k in gr_d:
        assert _close(gr_i[k], gr_d[k], 2e-5 if k != "att_va.weight" else 1e-4), k
    assert float(gr_i["att_va.weight"].abs().max()) > 0
    # repeated bank rows inside an image are the case the segmented sum exists for: make sure the batch has them
    flat = idx.reshape(cfg["B"], -1)
    assert any(len(torch.unique(r[r >= 0])) < int((r >= 0).sum()) for r in flat)
    print("index lists %d bytes per batch, dense tensor %d bytes" % (idx.numel() * 4, dense.numel() * 4))
    # twice the same step: bit-identical gradients (ordered segmented sum, no atomics)
    _, _, _, gr_2 = _xe(m, det, reg, caps, gts)
    for k in gr_i:
        assert torch.equal(gr_i[k], gr_2[k]), k


def test_xe_gradients_on_index_lists_vs_oracle_autograd():
    cfg = SMALL
    gains = {k: 1.0 for k in synth.DEFAULT_GAINS}
    w, det, idx, caps, gts = _inputs(cfg, 33, gains)
    m = helpers.build_model(cfg, w, DEV)
    reg = IndexedRegions(det.to(DEV), idx.to(DEV))
    _, _, loss, grads = _xe(m, det.to(DEV), reg, caps.to(DEV), gts.to(DEV))
    o = vo.Oracle(w, cfg["T"], 2, as_written=False)
    params = {k: o.p[k].requires_grad_(True) for k in o.p}
    out, gate = o.forward(det, caps, reg.dense().cpu())
    lo = vo.xe_loss(out, gate, caps, gts)[0]
    lo.backward()
    assert abs(loss - lo.item()) < 1e-4
    for k, p in params.items():
        assert _close(grads[k].cpu(), p.grad, 2e-3), k


def test_scst_step_on_index_lists_equals_dense():
    cfg = SMALL
    w, det, idx, _, _ = _inputs(cfg, 44)
    idx = torch.from_numpy(synth.make_slot_indices(cfg["B"], cfg["L"], cfg["R"], cfg["R0"], seed=44))
    m = helpers.build_model(cfg, w, DEV).train()
    det, idx = det.to(DEV), idx.to(DEV)
    reg = IndexedRegions(det, idx)
    dense = reg.dense().contiguous()
    with torch.no_grad():
        (sw, sg), _ = m.sample_rl(det, dense, seed=5)
    res = []
    for r in (dense, reg):
        m.zero_grad()
        (_, _), (lw, lg) = m.sample_rl(det, r, forced=(sw, sg))
        assert lw.requires_grad
        adv = torch.linspace(-1, 1, cfg["B"], device=DEV)
        vo.scst_loss(lw, lg, adv, torch.zeros_like(adv)).backward()
        res.append((lw.detach(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}))
    assert float((res[0][0] - res[1][0]).abs().max()) < 2e-5
    for k in res[0][1]:
        assert _close(res[1][1][k], res[0][1][k], 2e-5 if k != "att_va.weight" else 1e-4), k


def test_training_with_a_row_to_image_map_is_refused():
    cfg = SMALL
    w, det, idx, caps, gts = _inputs(cfg, 55)
    m = helpers.build_model(cfg, w, DEV).train()
    row_img = torch.arange(cfg["B"], dtype=torch.int32, device=DEV)
    reg = IndexedRegions(det.to(DEV), idx.to(DEV), row_img)
    with pytest.raises(RuntimeError, match="one decoder row per image"):
        m((det.to(DEV),), (caps.to(DEV), reg))
