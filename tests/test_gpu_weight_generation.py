"""The weights' generation contract (round-5 review, Missing #1 / Weak #1).

The reference reads its parameters live at every step (/root/reference/models/controllable_captioning.py:151-152,177-178) and its
training loop steps the optimizer between two forwards (/root/reference/coco_scripts/train.py:77,103,112-113).  The library keeps state
DERIVED from the weights (fp16-pair images in f16x2, bf16 copies in bf16, the decode cache in eval mode); this module pins that the
derived state follows the weights whoever moves them - in particular fused optimizers, which update the parameters in place WITHOUT
bumping Tensor._version on this torch build (the round-5 library keyed its refresh on that counter alone and trained on step-0 images).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
import helpers
import vsr_oracle as vo

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _fixture(name="g1_xe_small"):
    meta, _ = load_golden(name)
    cfg = meta["cfg"]
    w = helpers.weights_for(cfg, gains=meta["gains"])
    det, ctrl_seq, caps, gts = (x.to(DEV) for x in helpers.train_inputs(cfg, meta["seed"]))
    return cfg, w, det, ctrl_seq, caps, gts


def _train(m, make_opt, det, ctrl_seq, caps, gts, steps):
    m.train()
    opt = make_opt(m.parameters())
    losses = []
    for _ in range(steps):
        opt.zero_grad()
        out, gate = m((det,), (caps, ctrl_seq))
        loss = vo.xe_loss(out, gate, caps, gts)[0]
        loss.backward()
        opt.step()
        losses.append(loss.item())
    return losses


OPTS = {
    "adam": lambda **kw: (lambda ps: torch.optim.Adam(ps, lr=5e-4, **kw)),
    "sgd": lambda **kw: (lambda ps: torch.optim.SGD(ps, lr=2e-3, momentum=0.9, **kw)),
}


@pytest.mark.parametrize("opt", ["adam", "sgd"])
def test_fused_optimizer_trains_like_foreach_and_eval_sees_the_final_weights(opt):
    """3 XE steps with Adam / SGD(fused=True) == 3 steps with (foreach=True): same losses, same weights; then .eval() greedy and beam-3
    == a FRESH model loaded with the final state_dict.  Fails on the round-5 library (steps 2 and 3 ran on step-0 images)."""
    cfg, w, det, ctrl_seq, caps, gts = _fixture()
    ctrl = ctrl_seq[:, :cfg["L"]].contiguous() if ctrl_seq.size(1) >= cfg["L"] else ctrl_seq
    runs = {}
    for kind in ("fused", "foreach"):
        m = helpers.build_model(cfg, w, DEV)
        losses = _train(m, OPTS[opt](**{kind: True}), det, ctrl_seq, caps, gts, 3)
        runs[kind] = (m, losses)
    lf, le = runs["fused"][1], runs["foreach"][1]
    assert abs(lf[1] - lf[0]) > 1e-3 and abs(lf[2] - lf[1]) > 1e-3, "the loss must move: %s" % (lf,)
    np.testing.assert_allclose(lf, le, atol=2e-6, rtol=0)
    mf, me = runs["fused"][0], runs["foreach"][0]
    # the two runs see bit-identical gradients (the library is deterministic); what differs is the arithmetic of torch's fused and
    # foreach kernels, and Adam's m / (sqrt(v) + eps) amplifies their last-bit differences on near-zero gradients to ~1e-3 of one
    # update (observed: 1.0e-7 on weights of 0.09 at lr 5e-4): the bound is 2e-3 of lr per step on top of 1e-6 of the weight scale
    lr = 5e-4 if opt == "adam" else 2e-3
    for (k, a), (_, b) in zip(mf.named_parameters(), me.named_parameters()):
        scale = b.abs().max().item() + 1e-30
        assert (a - b).abs().max().item() <= 1e-6 * scale + 3 * 2e-3 * lr, k
    # evaluation after the fused run: derived state of the FINAL weights
    mf.eval()
    fresh = helpers.build_model(cfg, {k: v.detach().cpu().numpy() for k, v in mf.state_dict().items()}, DEV)
    with torch.no_grad():
        a = mf.test(det, ctrl)
        b = fresh.test(det, ctrl)
        (ba, _), (bb, _) = mf.beam_search((det, ctrl), [3, -1], 3, 1), fresh.beam_search((det, ctrl), [3, -1], 3, 1)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert torch.equal(ba[0], bb[0]) and torch.equal(ba[1], bb[1])


def test_fused_optimizer_follows_the_oracle_trajectory():
    """the same three fused steps against the CPU oracle under plain Adam (the existing visible-step test, with the optimizer bench.py uses)"""
    cfg, w, det, ctrl_seq, caps, gts = _fixture()
    m = helpers.build_model(cfg, w, DEV)
    losses = _train(m, OPTS["adam"](fused=True), det, ctrl_seq, caps, gts, 3)
    o = vo.Oracle(w, cfg["T"], 2, as_written=True)
    params = [o.p[k].requires_grad_(True) for k in o.p]
    oopt = torch.optim.Adam(params, lr=5e-4)
    ol = []
    for _ in range(3):
        oopt.zero_grad()
        oo, og = o.forward(det.cpu(), caps.cpu(), ctrl_seq.cpu())
        l = vo.xe_loss(oo, og, caps.cpu(), gts.cpu())[0]
        l.backward()
        oopt.step()
        ol.append(l.item())
    np.testing.assert_allclose(losses, ol, atol=2e-4, rtol=0)


def test_eval_between_fused_steps_sees_each_step():
    """an evaluation call between optimizer steps (model stays in eval mode the whole time, gradients taken through a second, training
    twin that SHARES the parameters): the process-wide optimizer-step count voids the eval model's decode cache and images"""
    cfg, w, det, ctrl_seq, caps, gts = _fixture()
    m = helpers.build_model(cfg, w, DEV)          # eval mode
    opt = torch.optim.SGD(m.parameters(), lr=0.5, fused=True)
    with torch.no_grad():
        out0, _ = m((det,), (caps, ctrl_seq))
    for p in m.parameters():
        p.grad = torch.ones_like(p) * 1e-2
    opt.step()                                     # fused: no _version bump
    with torch.no_grad():
        out1, _ = m((det,), (caps, ctrl_seq))
    fresh = helpers.build_model(cfg, {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}, DEV)
    with torch.no_grad():
        want, _ = fresh((det,), (caps, ctrl_seq))
    assert (out1 - out0).abs().max().item() > 1e-3, "the step must change the outputs"
    assert torch.equal(out1, want)


def test_p_data_edit_needs_invalidate_cache_and_gets_it():
    """writes nothing can see (p.data edits in eval mode): invalidate_cache() is the documented way, and it works"""
    cfg, w, det, ctrl_seq, caps, gts = _fixture()
    m = helpers.build_model(cfg, w, DEV)
    with torch.no_grad():
        m((det,), (caps, ctrl_seq))
    for p in m.parameters():
        p.data.mul_(1.05)
    m.invalidate_cache()
    with torch.no_grad():
        got, _ = m((det,), (caps, ctrl_seq))
    fresh = helpers.build_model(cfg, {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}, DEV)
    with torch.no_grad():
        want, _ = fresh((det,), (caps, ctrl_seq))
    assert torch.equal(got, want)


def test_bf16_mode_refreshes_its_copies_under_a_fused_optimizer(gemm_flavour):
    if gemm_flavour not in (None, "f16x2"):
        pytest.skip("picks its own compute dtype: run once")
    cfg, w, det, ctrl_seq, caps, gts = _fixture("g1_xe_wide")
    runs = []
    for kind in ("fused", "foreach"):
        m = helpers.build_model(cfg, w, DEV).set_compute_dtype("bf16")
        runs.append(_train(m, OPTS["adam"](**{kind: True}), det, ctrl_seq, caps, gts, 3))
    assert abs(runs[0][1] - runs[0][0]) > 1e-3 and abs(runs[0][2] - runs[0][1]) > 1e-3, "the loss must move: %s" % (runs[0],)
    np.testing.assert_allclose(runs[0], runs[1], atol=5e-6, rtol=2e-5)      # (1-ulp differences of the two Adam kernels, amplified by the bf16 rounding of the copies)


def test_inputs_created_under_inference_mode():
    """tensors made under torch.inference_mode() track no version counter: prepare() must not read it (it raised in round 5), and such
    inputs are never served from the prepare cache (an in-place rewrite would be invisible)"""
    cfg, w, det, ctrl_seq, caps, gts = _fixture()
    m = helpers.build_model(cfg, w, DEV)
    ctrl = ctrl_seq[:, :cfg["L"]].contiguous()
    with torch.no_grad():
        want = m.test(det, ctrl)
    with torch.inference_mode():
        d2, c2 = det.clone(), ctrl.clone()
        got = m.test(d2, c2)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
        c2.zero_()
        c2[:, :, 0] = ctrl[:, :, 0]                 # rewritten in place: must be prepared again
        got2 = m.test(d2, c2)
    with torch.no_grad():
        c3 = torch.zeros_like(ctrl)
        c3[:, :, 0] = ctrl[:, :, 0]
        want2 = m.test(det, c3)
    assert torch.equal(got2[0], want2[0]) and torch.equal(got2[1], want2[1])
