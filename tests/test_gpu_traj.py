"""Training TRAJECTORY against the reference (round-5 review, Missing #2): the loop of /root/reference/coco_scripts/train.py:92-120 -
model.train(); per batch forward (:103), the two NLL losses (:106-110), zero_grad / backward / Adam(lr=5e-4).step() (:77, :111-113) -
run for 5 steps on the imported reference at batch 100, full size (tests/golden/make_golden.py traj -> g14_xe_traj; a different
synthetic batch per step).  Every full-size training check before this one was a single step; a library that trains on stale weight
images (round 5 under fused optimizers) passes those and fails here from step 2 on.

The synthetic features are scaled by 2^-4 in this fixture (exact in fp32; recorded in the fixture): at scale 1 the first Adam step moves
the shift logit - a RAW sum of up to 36 region scores, controllable_captioning.py:187 - so far that the gate loss jumps 2.5 -> 80; at
2^-4 it still jumps 0.78 -> 18.5 -> 7.4 -> 0.81 -> 3.6, which is what makes the trajectory a sharp test.  The reference reproduces
itself to 3e-6 on it between 3 and 8 CPU threads."""
import numpy as np
import pytest
import torch

from conftest import load_golden
import helpers
import vsr_oracle as vo
from vsrcap import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _batch(cfg, seed, fs):
    det, ctrl_seq, caps, gts = helpers.train_inputs(cfg, seed)
    return (det * fs).to(DEV), (ctrl_seq * fs).to(DEV), caps.to(DEV), gts.to(DEV)


def _run(name, opt_kw, dtype=None):
    meta, g = load_golden(name)
    cfg, fs = meta["cfg"], meta["feat_scale"]
    w = helpers.weights_for(cfg, gains=meta["gains"])
    m = helpers.build_model(cfg, w, DEV)
    if dtype:
        m.set_compute_dtype(dtype)
    w0 = {k: p.detach().clone() for k, p in m.named_parameters()}
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=meta["lr"], **opt_kw)
    losses = []
    for i in range(meta["steps"]):
        det, ctrl_seq, caps, gts = _batch(cfg, meta["seed"] + i, fs)
        out, gate = m((det,), (caps, ctrl_seq))
        loss, lc, lg = vo.xe_loss(out, gate, caps, gts)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append([loss.item(), lc.item(), lg.item()])
    dn = np.array([float((p.detach() - w0[k]).double().norm()) for k, p in m.named_parameters()])
    m.eval()
    det, ctrl_seq, caps, gts = _batch(cfg, meta["seed"] + meta["steps"], fs)
    with torch.no_grad():
        out, gate = m((det,), (caps, ctrl_seq))
        held = [x.item() for x in vo.xe_loss(out, gate, caps, gts)]
    return meta, g, np.array(losses), dn, np.array(held)


@pytest.mark.parametrize("opt_kw", [dict(), dict(fused=True)], ids=["adam", "adam_fused"])
@pytest.mark.parametrize("name", ["g14_xe_traj_small", "g14_xe_traj"])
def test_five_xe_steps_follow_the_reference(name, opt_kw):
    meta, g, losses, dn, held = _run(name, opt_kw)
    # per step: total loss, caption loss, gate loss (the total carries 4 x the gate loss: train.py:110)
    np.testing.assert_allclose(losses[:, 1], g["losses"][:, 1], atol=1e-4, rtol=0)
    np.testing.assert_allclose(losses[:, 2], g["losses"][:, 2], atol=1e-4, rtol=1e-5)
    np.testing.assert_allclose(losses[:, 0], g["losses"][:, 0], atol=1e-4, rtol=2e-5)
    # where the weights went: norms of the 28 parameters' total change over the 5 steps
    np.testing.assert_allclose(dn, g["delta_norm"], rtol=2e-3, atol=1e-9)
    # and what an evaluation of the final weights sees (eval mode, no graph: decode-side derived state of the FINAL weights)
    np.testing.assert_allclose(held, g["heldout_losses"], atol=1e-4, rtol=2e-5)


def test_five_xe_steps_bf16_deviation_stated(gemm_flavour):
    """bf16 is not a parity mode: the trajectory's deviation from the reference is STATED (observed x 1.5), with the fused optimizer
    bench.py uses; a run on stale copies would be off by whole units from step 2 on"""
    if gemm_flavour not in (None, "f16x2"):
        pytest.skip("picks its own compute dtype: run once")
    meta, g, losses, dn, held = _run("g14_xe_traj", dict(fused=True), dtype="bf16")
    d = np.abs(losses - g["losses"])
    print("bf16 trajectory deviation per step (total, cap, gate):", d.tolist(), "held-out:", np.abs(held - g["heldout_losses"]).tolist())
    # observed on the box (round 6): caption loss <= 7.6e-6, gate loss <= 1.8e-2 absolute = 5.1e-3 of its value where it has jumped
    assert d[:, 1].max() < 5e-5                       # caption loss
    assert (d[:, 2] / np.maximum(1.0, g["losses"][:, 2])).max() < 1e-2       # gate loss, relative where it has jumped
    np.testing.assert_allclose(dn, g["delta_norm"], rtol=5e-2)
