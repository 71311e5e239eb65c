"""GPU tests of the index-list region format: vsr_prepare_indexed / vsr_reorder_slots through the
reference-shaped Python class.  Indexed decoding must give the tokens of the dense tensor the list stands for; the slot
re-ordering must reproduce the reference's own statements (golden g8) bit for bit."""
import numpy as np
import pytest
import torch

from conftest import load_golden
import helpers
import regions_oracle as ro
from test_regions import _cases
from vsrcap import evalbatch, regions, synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _small_model(table=None):
    meta, _ = load_golden("g3_beam_small")
    w = helpers.weights_for(meta["cfg"], wseed=meta.get("wseed", 0))
    return meta, helpers.build_model(meta["cfg"], w, DEV, bos=meta["bos"], verb_table=table or meta["verb_table"]), w


def _indexed_inputs(cfg, n_img, caps_per_img, seed, Rb=None):
    Rb = Rb or cfg["R0"]
    det = torch.from_numpy(synth.make_detections(n_img, cfg["R0"], cfg["D"], seed=seed)).to(DEV)
    bank = det if Rb == cfg["R0"] else torch.from_numpy(synth.make_detections(n_img, Rb, cfg["D"], seed=seed + 50)).to(DEV)
    N = n_img * caps_per_img
    idx = torch.from_numpy(synth.make_slot_indices(N, cfg["L"], cfg["R"], Rb, seed=seed)).to(DEV)
    row_img = torch.arange(N, dtype=torch.int32, device=DEV) // caps_per_img if caps_per_img > 1 else None
    return det, regions.IndexedRegions(bank, idx, row_img)


def test_reorder_slots_matches_reference_golden():
    _, m, _ = _small_model()
    eng = m._engine(torch.device(DEV))
    for c in _cases("g8_reorder"):
        L = c["idx"].shape[0]
        bank = torch.from_numpy(c["bank"]).to(DEV)[None]
        reg = regions.IndexedRegions(bank, torch.from_numpy(c["idx"]).to(DEV)[None])
        out, verbs = regions.reorder_slots(eng, reg, [list(c["rank"])], c["verbs"][None])
        dense = ro.gather_dense(c["bank"], out.slot_idx[0].cpu().numpy())
        np.testing.assert_array_equal(dense.astype(np.float64), c["recons"])
        np.testing.assert_array_equal(verbs[0].cpu().numpy().astype(np.float64), c["verbs_out"][:, 0])
        np.testing.assert_array_equal(out.dense()[0].cpu().numpy().astype(np.float64), c["recons"])   # device-side gather too


def test_reorder_slots_batched_with_shared_banks():
    """all captions of several images in ONE call, banks shared through row_img"""
    meta, m, _ = _small_model()
    cfg = meta["cfg"]
    eng = m._engine(torch.device(DEV))
    det, reg = _indexed_inputs(cfg, 3, 4, seed=5)
    N, L = reg.slot_idx.shape[:2]
    rng = np.random.RandomState(0)
    ranks = [list(rng.permutation(L))[:rng.randint(1, L + 1)] for _ in range(N)]
    verbs = np.where(rng.rand(N, L) > 0.5, rng.randint(0, 9, size=(N, L)), -1).astype(np.float64)
    out, vout = regions.reorder_slots(eng, reg, ranks, verbs)
    dense_in = reg.dense().cpu().numpy().astype(np.float64)
    n_ok = 0
    for n in range(N):
        try:
            row, vrow = ro.reconstruct_dense(dense_in[n], ranks[n], verbs[n][:, None], L)
        except ValueError:          # every ranked slot is empty: the reference's :234 cannot broadcast; here: all padding
            assert (out.slot_idx[n] == -1).all()
            continue
        n_ok += 1
        np.testing.assert_array_equal(out.dense()[n].cpu().numpy().astype(np.float64), row)
        np.testing.assert_array_equal(vout[n].cpu().numpy().astype(np.float64), vrow[:, 0])
    assert n_ok >= N // 2
    with pytest.raises(IndexError):
        regions.reorder_slots(eng, reg, [[L]] * N, verbs)


@pytest.mark.parametrize("caps_per_img,Rb", [(1, None), (3, None), (2, 14)])
def test_indexed_decode_equals_dense_decode(caps_per_img, Rb):
    import vsr_oracle as vo
    meta, m, w = _small_model()
    cfg = meta["cfg"]
    det, reg = _indexed_inputs(cfg, 4, caps_per_img, seed=21, Rb=Rb)
    N = reg.slot_idx.size(0)
    dense = reg.dense().contiguous()
    det_rows = det if reg.row_img is None else det[reg.row_img.long()].contiguous()
    verbs = torch.from_numpy(synth.make_verbs(N, cfg["L"], meta["nv"], seed=3, p=0.3)).to(DEV)
    with torch.no_grad():
        wi, gi = m.test(det, reg)
        wd, gd = m.test(det_rows, dense)
        assert wi.shape == (N, cfg["T"])
        np.testing.assert_array_equal(wi.cpu().numpy(), wd.cpu().numpy())
        np.testing.assert_array_equal(gi.cpu().numpy(), gd.cpu().numpy())
        (bwi, bgi), _ = m.beam_search((det, reg), meta["eos"], 5, 1)
        (bwd, bgd), _ = m.beam_search((det_rows, dense), meta["eos"], 5, 1)
        np.testing.assert_array_equal(bwi.cpu().numpy(), bwd.cpu().numpy())
        np.testing.assert_array_equal(bgi.cpu().numpy(), bgd.cpu().numpy())
        for gt in (False, True):
            (vwi, vgi), _ = m.beam_search_v((det, reg, verbs), eos_idxs=meta["eos"], beam_size=3, out_size=1, gt=gt)
            (vwd, vgd), _ = m.beam_search_v((det_rows, dense, verbs), eos_idxs=meta["eos"], beam_size=3, out_size=1, gt=gt)
            np.testing.assert_array_equal(vwi.cpu().numpy(), vwd.cpu().numpy())
            np.testing.assert_array_equal(vgi.cpu().numpy(), vgd.cpu().numpy())
        # teacher-forced log-probs through the indexed path (L == T slots) against the CPU oracle on the dense tensor
        caps = torch.from_numpy(synth.make_captions(N, cfg["L"], cfg["V"], seed=9)).to(DEV)
        out_i, gate_i = m((det,), (caps, reg))
    o = vo.Oracle(w, cfg["T"], meta["bos"], as_written=False)
    with torch.no_grad():
        ow, og = o.test(det_rows.cpu(), dense.cpu())
        oo, ogate = o.forward(det_rows.cpu(), caps.cpu(), dense.cpu())
    np.testing.assert_array_equal(wi.cpu().numpy(), ow.numpy())
    np.testing.assert_array_equal(gi.cpu().numpy(), og.numpy())
    np.testing.assert_allclose(gate_i.cpu().numpy(), ogate.numpy(), atol=2e-4, rtol=0)
    np.testing.assert_allclose(out_i.cpu().numpy(), oo.numpy(), atol=2e-4, rtol=0)


def test_indexed_eval_batch_equals_per_image_reference_flow():
    """evalbatch.beam_search_v_indexed (one call for the batch) == the reference's per-image flow on dense tensors:
    reconstruct (oracle restatement of eval_coco.py:222-238) -> expand det -> beam_search_v per image."""
    meta, m, _ = _small_model()
    cfg = meta["cfg"]
    n_img, caps = 3, 3
    det, reg = _indexed_inputs(cfg, n_img, caps, seed=33)
    N, L = reg.slot_idx.shape[:2]
    rng = np.random.RandomState(4)
    ranks = [list(rng.permutation(L))[:rng.randint(2, L + 1)] for _ in range(N)]
    verb_list = np.where(rng.rand(N, L) > 0.7, rng.randint(0, meta["nv"], size=(N, L)), -1).astype(np.float64)
    with torch.no_grad():
        (w_all, g_all), _ = evalbatch.beam_search_v_indexed(m, det, reg.bank, reg.slot_idx, reg.row_img, ranks, verb_list,
                                                          meta["eos"], beam_size=5, out_size=1, gt=False)
    dense_in = reg.dense().cpu().numpy().astype(np.float64)
    for i in range(n_img):
        rows, vrows = [], []
        for n in range(i * caps, (i + 1) * caps):
            row, vrow = ro.reconstruct_dense(dense_in[n], ranks[n], verb_list[n][:, None], L)
            rows.append(row)
            vrows.append(vrow[:, 0])
        recons = torch.tensor(np.stack(rows)).float().to(DEV)                   # eval_coco.py:241
        verbs_i = torch.tensor(np.stack(vrows)).to(DEV)
        det_i = det[i].unsqueeze(0).expand(caps, det.size(1), det.size(2))      # :242
        with torch.no_grad():
            (w_i, g_i), _ = m.beam_search_v((det_i, recons, verbs_i), eos_idxs=meta["eos"], beam_size=5, out_size=1, gt=False)
        np.testing.assert_array_equal(w_all[i * caps:(i + 1) * caps].cpu().numpy(), w_i.cpu().numpy())
        np.testing.assert_array_equal(g_all[i * caps:(i + 1) * caps].cpu().numpy(), g_i.cpu().numpy())


def test_indexed_errors_are_loud():
    meta, m, _ = _small_model()
    cfg = meta["cfg"]
    det, reg = _indexed_inputs(cfg, 2, 1, seed=2)
    bad = reg.slot_idx.clone()
    bad[0, 0, 0] = cfg["R0"]                                   # one past the bank
    with torch.no_grad(), pytest.raises(RuntimeError, match="outside the feature bank"):
        m.test(det, regions.IndexedRegions(reg.bank, bad))
    with torch.no_grad(), pytest.raises(RuntimeError, match="row_img"):
        m.test(det, regions.IndexedRegions(reg.bank, torch.cat([reg.slot_idx, reg.slot_idx], 0)))
    caps = torch.from_numpy(synth.make_captions(2, cfg["L"], cfg["V"], seed=1)).to(DEV)
    m.train()
    if reg.row_img is None:                                    # one row per image: index lists train (tests/test_gpu_train_indexed.py)
        out, _ = m((det,), (caps, reg))
        assert out.requires_grad
    with pytest.raises(RuntimeError, match="one decoder row per image"):
        m((det,), (caps, regions.IndexedRegions(reg.bank, reg.slot_idx, torch.arange(reg.slot_idx.size(0), dtype=torch.int32, device=DEV))))
    m.eval()
    with torch.no_grad():                                      # the handle is usable again after the failures
        w1, _ = m.test(det, reg)
        w2, _ = m.test(det, reg.dense())
    np.testing.assert_array_equal(w1.cpu().numpy(), w2.cpu().numpy())
