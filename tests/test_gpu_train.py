"""GPU parity of the training path: the hand-written BPTT backward against (1) the reference's autograd gradient
norms captured in the golden fixtures and (2) the CPU oracle differentiated by torch autograd, element by element."""
import numpy as np
import pytest
import torch

from conftest import load_golden
import helpers
import vsr_oracle as vo
from vsrcap import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _oracle_grads(w, cfg, det, caps, ctrl_seq, gts, **flags):
    o = vo.Oracle(w, cfg["T"], 2, as_written=True, **flags)
    for k in o.p:
        o.p[k].requires_grad_(True)
    out, gate = o.forward(det, caps, ctrl_seq)
    loss = vo.xe_loss(out, gate, caps, gts)[0]
    loss.backward()
    return loss.item(), {k: o.p[k].grad for k in o.p}


def _device_grads(m, det, caps, ctrl_seq, gts):
    m.train()
    m.zero_grad()
    out, gate = m((det.to(DEV),), (caps.to(DEV), ctrl_seq.to(DEV)))
    assert out.requires_grad and gate.requires_grad
    loss = vo.xe_loss(out, gate, caps.to(DEV), gts.to(DEV))[0]
    loss.backward()
    return loss.item(), {k: p.grad.detach().cpu() for k, p in m.named_parameters()}


def _check(got, want, rtol):
    for k in want:
        g, r = got[k].double(), want[k].double()
        scale = r.abs().max().item() + 1e-12
        err = (g - r).abs().max().item()
        assert err <= rtol * scale + 1e-9, "%s: max err %.3e vs scale %.3e" % (k, err, scale)


@pytest.mark.parametrize("name", ["g1_xe_small", "g1_xe_hot_small"])
def test_xe_gradients_elementwise_vs_oracle_and_reference_norms(name):
    meta, g = load_golden(name)
    cfg = meta["cfg"]
    w = helpers.weights_for(cfg, gains=meta["gains"])
    det, ctrl_seq, caps, gts = helpers.train_inputs(cfg, meta["seed"])
    m = helpers.build_model(cfg, w, DEV)
    loss, got = _device_grads(m, det, caps, ctrl_seq, gts)
    ref_loss, want = _oracle_grads(w, cfg, det, caps, ctrl_seq, gts)
    assert abs(loss - ref_loss) < (1e-4 if "hot" not in name else 2e-3)
    _check(got, want, 2e-3)
    gn = np.array([float(got[k].double().norm()) for k in meta["param_order"]])
    np.testing.assert_allclose(gn, g["grad_norm"], rtol=2e-3, atol=1e-8)
    gs = np.array([float(got[k].double().sum()) for k in meta["param_order"]])
    np.testing.assert_allclose(gs, g["grad_sum"], rtol=5e-3, atol=2e-5)


@pytest.mark.parametrize("name", ["g1_xe_wide", "g1_xe_full"])
def test_xe_gradient_norms_vs_reference(name):
    meta, g = load_golden(name)
    cfg = meta["cfg"]
    w = helpers.weights_for(cfg, gains=meta["gains"])
    det, ctrl_seq, caps, gts = helpers.train_inputs(cfg, meta["seed"])
    m = helpers.build_model(cfg, w, DEV)
    loss, got = _device_grads(m, det, caps, ctrl_seq, gts)
    assert abs(loss - g["losses"][0]) < 1e-4
    gn = np.array([float(got[k].double().norm()) for k in meta["param_order"]])
    np.testing.assert_allclose(gn, g["grad_norm"], rtol=2e-3, atol=1e-8)


@pytest.mark.parametrize("T", [62, 63, 70])
def test_xe_gradients_long_sequences_across_the_dynamic_bound_table_limit(T):
    """The f16x2 backward pass keeps the measured bounds of its gradient operands in 8 slots per timestep of a 512-slot table: up to
    T = 62 it runs on the f16x2 kernels, from T = 63 on the f32x3 kernels (the forward pass stays f16x2; include/vsrcap.h,
    vsr_refresh_h2_weights).  Both sides of the limit against the oracle's autograd, element by element (round-4 advisor finding 4)."""
    cfg = dict(V=50, B=3, R0=6, R=6, D=64, L=T, T=T, E=64, H=64, A=32)
    w = helpers.weights_for(cfg, gains={k: 1.0 for k in synth.DEFAULT_GAINS})
    det, ctrl_seq, caps, gts = helpers.train_inputs(cfg, 23)
    m = helpers.build_model(cfg, w, DEV)
    loss, got = _device_grads(m, det, caps, ctrl_seq, gts)
    ref_loss, want = _oracle_grads(w, cfg, det, caps, ctrl_seq, gts)
    assert abs(loss - ref_loss) < 1e-4
    _check(got, want, 2e-3)


@pytest.mark.parametrize("flags", [dict(h2_first_lstm=False), dict(img_second_lstm=True)])
def test_xe_gradients_config_flags(flags):
    cfg = dict(V=61, B=5, R0=6, R=7, D=128, L=4, T=7, E=32, H=48, A=16)
    w = synth.make_weights(cfg["V"], cfg["D"], cfg["E"], cfg["H"], cfg["A"], seed=4, gains={k: 1.5 for k in synth.DEFAULT_GAINS}, **flags)
    det, ctrl_seq, caps, gts = helpers.train_inputs(cfg, 12)
    m = helpers.build_model(cfg, w, DEV, **flags)
    loss, got = _device_grads(m, det, caps, ctrl_seq, gts)
    ref_loss, want = _oracle_grads(w, cfg, det, caps, ctrl_seq, gts, **flags)
    assert abs(loss - ref_loss) < 1e-4
    _check(got, want, 2e-3)


def test_scst_step_gradients_vs_oracle():
    """train.py:151-178: sample_rl log-probs -> -(mean lp_w + mean lp_g) * (r - r_b) -> backward; rewards are inputs."""
    meta, _ = load_golden("g3_beam_small")
    cfg = meta["cfg"]
    w = helpers.weights_for(cfg)
    det, ctrl = helpers.decode_inputs(cfg, meta["seed"])
    m = helpers.build_model(cfg, w, DEV)
    m.train()
    m.zero_grad()
    (sw, sg), (lw, lg) = m.sample_rl(det.to(DEV), ctrl.to(DEV), seed=11)
    assert lw.requires_grad and lg.requires_grad
    reward = torch.from_numpy(synth.hash_u01(cfg["B"], 70, 1).astype(np.float32))
    base = torch.from_numpy(synth.hash_u01(cfg["B"], 71, 1).astype(np.float32))
    loss = vo.scst_loss(lw, lg, reward.to(DEV), base.to(DEV))
    loss.backward()
    got = {k: p.grad.detach().cpu() for k, p in m.named_parameters()}
    o = vo.Oracle(w, cfg["T"], 2, as_written=True)
    for k in o.p:
        o.p[k].requires_grad_(True)
    _, (olw, olg) = o.sample_rl(det, ctrl, forced=(sw.cpu(), sg.cpu()))
    oloss = vo.scst_loss(olw, olg, reward, base)
    oloss.backward()
    assert abs(loss.item() - oloss.item()) < 1e-4
    np.testing.assert_allclose(lw.detach().cpu().numpy(), olw.detach().numpy(), atol=2e-4, rtol=0)
    _check(got, {k: o.p[k].grad for k in o.p}, 3e-3)


def test_optimizer_step_stays_visible_to_the_library():
    """weights are borrowed, not copied: after an Adam step the next forward must use the new values."""
    meta, _ = load_golden("g1_xe_small")
    cfg = meta["cfg"]
    w = helpers.weights_for(cfg, gains=meta["gains"])
    det, ctrl_seq, caps, gts = helpers.train_inputs(cfg, meta["seed"])
    m = helpers.build_model(cfg, w, DEV)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=5e-4)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        out, gate = m((det.to(DEV),), (caps.to(DEV), ctrl_seq.to(DEV)))
        loss = vo.xe_loss(out, gate, caps.to(DEV), gts.to(DEV))[0]
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[2] < losses[1] < losses[0]
    # the same three steps on the oracle
    o = vo.Oracle(w, cfg["T"], 2, as_written=True)
    params = [o.p[k].requires_grad_(True) for k in o.p]
    oopt = torch.optim.Adam(params, lr=5e-4)
    ol = []
    for _ in range(3):
        oopt.zero_grad()
        oo, og = o.forward(det, caps, ctrl_seq)
        l = vo.xe_loss(oo, og, caps, gts)[0]
        l.backward()
        oopt.step()
        ol.append(l.item())
    np.testing.assert_allclose(losses, ol, atol=2e-4, rtol=0)


def test_bucketed_exchange_is_ordered_behind_the_gradients_it_carries(monkeypatch):
    """The overlapped exchange of DataParallelStep on ONE GPU (round-2 advisor finding): every gradient bucket is handed to an
    ASYNCHRONOUS device operation on the side stream behind the library's bucket event (vsr_train_wait_bucket).  With "all-reduce"
    = multiply by 2 (a stand-in with an exactly known result), every one of the 28 gradients must come out as exactly twice the
    gradient of a step without exchange: a bucket exchanged before one of its gradients was complete would leave that gradient
    undoubled (or half-written); so would a gradient assigned to the wrong bucket.  Also: the same through the bf16 wire format."""
    from vsrcap import parallel
    meta, _ = load_golden("g1_xe_wide")
    cfg = meta["cfg"]
    w = helpers.weights_for(cfg, gains=meta["gains"])
    det, ctrl_seq, caps, gts = (x.to(DEV) for x in helpers.train_inputs(cfg, meta["seed"]))
    m = helpers.build_model(cfg, w, DEV).train()
    opt = torch.optim.SGD(m.parameters(), lr=0.0)                    # lr 0: the gradients stay inspectable, the weights fixed

    def grads_of(step):
        step.xe_step(det, caps, ctrl_seq, gts)
        torch.cuda.synchronize()
        return [v.clone() for v in step.grads.views]

    plain = parallel.DataParallelStep(m, opt, forward_fn=lambda d, c, s: m((d,), (c, s)))
    ref = grads_of(plain)
    plain.close()
    calls = []

    def doubling(buf):                                               # asynchronous device work on the (side) stream it is called on
        calls.append(buf.numel())
        buf.mul_(2.0)
    for dt in (torch.float32, torch.bfloat16):
        calls.clear()
        st = parallel.DataParallelStep(m, opt, forward_fn=lambda d, c, s: m((d,), (c, s)), all_reduce_fn=doubling, exchange_dtype=dt)
        monkeypatch.setattr(st, "_world", lambda: 2)
        monkeypatch.setattr(st, "_sum", lambda t: t)                 # the loss statistics need no second rank here
        got = grads_of(st)
        assert len(calls) == st.n_buckets and sum(calls) == st.grads.flat.numel()
        for i, (a, b) in enumerate(zip(got, ref)):
            want = 2.0 * (b.bfloat16().float() if dt == torch.bfloat16 else b)
            assert torch.equal(a, want), "gradient %d (bucket order) differs from twice the plain gradient" % i
        st.close()
