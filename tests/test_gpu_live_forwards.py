"""More than one live forward per model (round-5 review, Missing #3).  The reference runs under eager autograd, where two forwards
followed by (l1 + l2).backward(), micro-batch accumulation, or a decode call between a forward and its backward just work
(/root/reference/models/CaptioningModel.py:22-36).  Rounds 1-5 kept ONE saved forward per handle and raised at the second backward; now
every live forward holds its own pair of workspaces (vsrcap/engine.py: _Slot; include/vsrcap.h: vsr_train_select)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
import helpers
import vsr_oracle as vo

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _setup():
    meta, _ = load_golden("g1_xe_small")
    cfg = meta["cfg"]
    w = helpers.weights_for(cfg, gains=meta["gains"])
    b1 = helpers.train_inputs(cfg, meta["seed"])
    b2 = helpers.train_inputs(cfg, meta["seed"] + 17)
    b2 = (b2[0] * 1.75, b2[1] * 1.75, b2[2], b2[3])      # other feature bounds: the f16x2 exponent rows of the two batches differ
    return cfg, w, b1, b2


def _oracle_sum_grads(cfg, w, batches):
    o = vo.Oracle(w, cfg["T"], 2, as_written=True)
    for k in o.p:
        o.p[k].requires_grad_(True)
    total = 0
    for det, ctrl_seq, caps, gts in batches:
        out, gate = o.forward(det, caps, ctrl_seq)
        total = total + vo.xe_loss(out, gate, caps, gts)[0]
    total.backward()
    return total.item(), {k: o.p[k].grad for k in o.p}


def _dev(b):
    return tuple(x.to(DEV) for x in b)


def _check(m, want, rtol=2e-3):
    for k, p in m.named_parameters():
        r = want[k].double()
        scale = r.abs().max().item() + 1e-12
        err = (p.grad.detach().cpu().double() - r).abs().max().item()
        assert err <= rtol * scale + 1e-9, "%s: max err %.3e vs scale %.3e" % (k, err, scale)


def test_two_forwards_then_one_backward_of_the_sum():
    cfg, w, b1, b2 = _setup()
    want_loss, want = _oracle_sum_grads(cfg, w, [b1, b2])
    m = helpers.build_model(cfg, w, DEV).train()
    m.zero_grad()
    d1, s1, c1, g1 = _dev(b1)
    d2, s2, c2, g2 = _dev(b2)
    out1, gate1 = m((d1,), (c1, s1))
    out2, gate2 = m((d2,), (c2, s2))
    assert m._engine(torch.device(DEV)).live_forwards() == 2
    loss = vo.xe_loss(out1, gate1, c1, g1)[0] + vo.xe_loss(out2, gate2, c2, g2)[0]
    loss.backward()
    assert abs(loss.item() - want_loss) < 2e-4
    _check(m, want)


@pytest.mark.parametrize("order", ["first_then_second", "second_then_first"])
def test_two_forwards_two_backwards_accumulate(order):
    cfg, w, b1, b2 = _setup()
    _, want = _oracle_sum_grads(cfg, w, [b1, b2])
    m = helpers.build_model(cfg, w, DEV).train()
    m.zero_grad()
    d1, s1, c1, g1 = _dev(b1)
    d2, s2, c2, g2 = _dev(b2)
    out1, gate1 = m((d1,), (c1, s1))
    out2, gate2 = m((d2,), (c2, s2))
    l1 = vo.xe_loss(out1, gate1, c1, g1)[0]
    l2 = vo.xe_loss(out2, gate2, c2, g2)[0]
    for l in ((l1, l2) if order == "first_then_second" else (l2, l1)):
        l.backward()
    _check(m, want)


def test_micro_batches_reuse_one_pair_of_buffers():
    """forward / backward pairs (gradient accumulation): the slot of a differentiated forward is reused, no second pair is opened"""
    cfg, w, b1, b2 = _setup()
    _, want = _oracle_sum_grads(cfg, w, [b1, b2])
    m = helpers.build_model(cfg, w, DEV).train()
    m.zero_grad()
    keep = []
    for b in (b1, b2):
        d, s, c, g = _dev(b)
        out, gate = m((d,), (c, s))
        l = vo.xe_loss(out, gate, c, g)[0]
        l.backward()
        keep.append((out, gate, l))                 # the graphs stay referenced, as in a loop that logs its losses later
    assert len(m._engine(torch.device(DEV))._slots) == 1
    _check(m, want)


def test_decode_between_a_forward_and_its_backward():
    """the greedy baseline of a self-critical step taken AFTER the forward whose gradient is still to come (train.py:127-138 / :151)"""
    cfg, w, b1, _ = _setup()
    _, want = _oracle_sum_grads(cfg, w, [b1])
    m = helpers.build_model(cfg, w, DEV).train()
    m.zero_grad()
    d, s, c, g = _dev(b1)
    out, gate = m((d,), (c, s))
    ctrl = s[:, :cfg["L"]].contiguous()
    with torch.no_grad():
        m.eval()
        words, _ = m.test(d, ctrl)
        (bw, _), _ = m.beam_search((d, ctrl), [3, -1], 3, 1)
        m.train()
    vo.xe_loss(out, gate, c, g)[0].backward()
    _check(m, want)
    fresh = helpers.build_model(cfg, w, DEV)
    with torch.no_grad():
        w2, _ = fresh.test(d, ctrl)
    assert torch.equal(words, w2)


def test_too_many_live_forwards_fail_loudly_at_the_forward():
    from vsrcap import engine
    cfg, w, b1, _ = _setup()
    m = helpers.build_model(cfg, w, DEV).train()
    d, s, c, g = _dev(b1)
    alive = [m((d,), (c, s)) for _ in range(engine.MAX_LIVE_FORWARDS)]
    with pytest.raises(RuntimeError, match="alive and not yet differentiated"):
        m((d,), (c, s))
    del alive                                        # dropping the outputs frees the buffers: the next forward goes through
    out, gate = m((d,), (c, s))
    vo.xe_loss(out, gate, c, g)[0].backward()


def test_second_backward_after_a_later_forward_raises():
    """retain_graph=True across a LATER forward: the differentiated forward's buffers were reused (stated limit, loud)"""
    cfg, w, b1, b2 = _setup()
    m = helpers.build_model(cfg, w, DEV).train()
    d1, s1, c1, g1 = _dev(b1)
    d2, s2, c2, g2 = _dev(b2)
    out1, gate1 = m((d1,), (c1, s1))
    l1 = vo.xe_loss(out1, gate1, c1, g1)[0]
    l1.backward(retain_graph=True)
    l1.backward(retain_graph=True)                   # same forward, buffers untouched: fine
    out2, gate2 = m((d2,), (c2, s2))                 # reuses the differentiated forward's buffers
    with pytest.raises(RuntimeError, match="saved activations are gone"):
        l1.backward()
    vo.xe_loss(out2, gate2, c2, g2)[0].backward()
