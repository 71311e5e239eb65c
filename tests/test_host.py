"""CPU checks of the host-side mirror: constructor, state_dict contract, loud failure without a GPU."""
import pytest
import torch

import helpers
from vsrcap import synth


def test_state_dict_keys_and_shapes_match_reference_contract():
    from models import ControllableCaptioningModel
    for flags in (dict(), dict(h2_first_lstm=False), dict(img_second_lstm=True)):
        m = ControllableCaptioningModel(20, 123, 2, det_feat_size=64, input_encoding_size=32, rnn_size=40, att_size=16,
                                        verb_2_vob_all={}, **flags)
        want = synth.param_shapes(123, 64, 32, 40, 16, **flags)
        got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        assert list(got.keys()) == list(want.keys())
        assert got == {k: tuple(v) for k, v in want.items()}
    # zero biases, orthogonal recurrent weights (init_weights :72-107)
    assert float(m.out_fc.bias.abs().sum()) == 0 and float(m.lstm_cell_1.bias_ih.abs().sum()) == 0
    w = m.lstm_cell_1.weight_hh
    assert torch.allclose(w.t() @ w, torch.eye(40), atol=1e-4)


def test_default_size_parameter_count():
    n = sum(int(torch.tensor(s).prod()) for s in synth.param_shapes(10000).values())
    assert n == 71146160


def test_ctor_reads_json_tables_from_cwd(tmp_path, monkeypatch):
    from models import ControllableCaptioningModel
    monkeypatch.chdir(tmp_path)
    with pytest.raises(FileNotFoundError):
        ControllableCaptioningModel(5, 20, 2, det_feat_size=8, input_encoding_size=8, rnn_size=8, att_size=8)
    (tmp_path / "datasets" / "coco").mkdir(parents=True)
    (tmp_path / "datasets" / "coco" / "verb_2_vob_all_refine.json").write_text('{"3": [4, 5]}')
    (tmp_path / "datasets" / "coco" / "verb_2_vob.json").write_text('{}')
    m = ControllableCaptioningModel(5, 20, 2, det_feat_size=8, input_encoding_size=8, rnn_size=8, att_size=8)
    assert m.verb_2_vob_all == {"3": [4, 5]}


def test_compute_on_cpu_fails_loudly():
    cfg = dict(V=20, B=2, R0=3, R=3, D=8, L=2, T=4, E=8, H=8, A=8)
    m = helpers.build_model(cfg, synth.make_weights(20, 8, 8, 8, 8), "cpu")
    det, ctrl = helpers.decode_inputs(cfg, 1)
    with pytest.raises(RuntimeError, match="GPU"):
        m.test(det, ctrl)
    with pytest.raises(RuntimeError, match="GPU"):
        m.beam_search((det, ctrl), [3, -1], 2, 1)
    st = m.init_state(2, torch.device("cpu"))
    assert st[0][0].shape == (2, 8) and st[2].dtype == torch.long


def test_synth_is_deterministic_and_padded():
    a = synth.make_ctrl(3, 4, 6, 256, seed=9)
    b = synth.make_ctrl(3, 4, 6, 256, seed=9)
    assert (a == b).all() and (a >= 0).all()
    nz = (a.sum(-1) != 0)
    assert nz[:, :, 0].all()                      # at least one valid row per slot
    assert ((nz[:, :, 1:] <= nz[:, :, :-1])).all()  # valid rows first, zero padding after
    d = synth.make_detections(5, 12, 256, seed=9)
    assert ((d.sum(-1) != 0).sum(1) >= 1).all()


def test_empty_batch_follows_the_reference():
    """reference probed here (CPU, torch 2.10): test() on a batch of 0 images returns (0, T) int64 outputs; beam_search and
    sample_rl raise RuntimeError (a reshape of 0 elements) - ours raise RuntimeError too (no kernel is launched for B = 0)."""
    from models import ControllableCaptioningModel
    m = ControllableCaptioningModel(6, 30, 2, det_feat_size=16, input_encoding_size=8, rnn_size=8, att_size=4, verb_2_vob_all={})
    det, ctrl = torch.rand(0, 5, 16), torch.rand(0, 3, 5, 16)
    w, g = m.test(det, ctrl)
    assert w.shape == (0, 6) and g.shape == (0, 6) and w.dtype == torch.int64 and g.dtype == torch.int64
    with pytest.raises(RuntimeError):
        m.beam_search((det, ctrl), [3, -1], 3, 1)
    with pytest.raises(RuntimeError):
        m.sample_rl(det, ctrl)


def test_weights_generation_moves_with_everything_that_can_move_the_weights():
    """The derived weight state (fp16-pair images, bf16 copies, decode cache) is keyed on model._weights_version(): it must change on an
    optimizer step of ANY torch optimizer anywhere (global post-step hook - fused optimizers do not bump Tensor._version), on train() /
    eval() switches, load_state_dict(), invalidate_cache() and an attached optimizer's step; and stay put otherwise.  (CPU: bookkeeping only.)"""
    from models import ControllableCaptioningModel
    m = ControllableCaptioningModel(5, 20, 2, det_feat_size=8, input_encoding_size=8, rnn_size=8, att_size=8, verb_2_vob_all={})
    v0 = m._weights_version()
    assert m._weights_version() == v0                        # reading it changes nothing
    m.eval()
    v1 = m._weights_version()
    assert v1 != v0
    m.eval()
    assert m._weights_version() == v1                        # no switch, no bump
    m.train()
    v2 = m._weights_version()
    assert v2 != v1
    # an optimizer that has nothing to do with this model: the process-wide step count still moves (conservative by design)
    other = torch.nn.Parameter(torch.zeros(3))
    other.grad = torch.ones(3)
    torch.optim.SGD([other], lr=0.1).step()
    v3 = m._weights_version()
    assert v3 != v2
    # this model's optimizer
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    for p in m.parameters():
        p.grad = torch.zeros_like(p)
    opt.step()
    v4 = m._weights_version()
    assert v4 != v3
    m.load_state_dict(m.state_dict())
    v5 = m._weights_version()
    assert v5 != v4
    m.invalidate_cache()
    assert m._weights_version() != v5

    class Bare:                                              # an "optimizer" without torch's hooks: attach_optimizer has nothing to register
        pass
    assert m.attach_optimizer(Bare()) is None
    h = m.attach_optimizer(opt)
    v6 = m._weights_version()
    opt.step()
    assert m._weights_version()[0] > v6[0]                   # the attached hook bumped the model's own counter
    h.remove()


def test_live_forward_slots_bookkeeping():
    """vsrcap.engine._Slot: a slot is 'live' while its forward's token is alive and the forward has not been differentiated"""
    import gc
    from vsrcap.engine import _Slot, ForwardToken
    import weakref
    s = _Slot()
    assert not s.live()
    tok = ForwardToken(7)
    s.token, s.generation, s.differentiated = weakref.ref(tok), 7, False
    assert s.live()
    s.differentiated = True
    assert not s.live()
    s.differentiated = False
    del tok
    gc.collect()
    assert not s.live()
