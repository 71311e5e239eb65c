"""Autograd glue of the training path: ONE autograd.Function whose forward is vsr_train_forward (teacher-forced or
sample-replayed unroll, activations saved by the library) and whose backward is the hand-written BPTT
vsr_train_backward.  torch only sees (out, gate) -> 28 parameter gradients; nothing is differentiated by torch.

Callers reproduced:
  coco_scripts/train.py:103-113   out, gate = model((det,), (captions, ctrl_det_seqs)); NLL losses; loss.backward()
  coco_scripts/train.py:151-178   outs, log_probs = model.sample_rl(det, ctrl); loss = -(mean lp_w + mean lp_g) * (r - r_b)
"""
import torch

from . import _lib

_PARAM_KEYS = [k for _, k in _lib.WEIGHT_FIELDS]


class _DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, B, device, word_in, slots, *params):
        out, gate = eng.train_forward(B, device, word_in, slots)
        ctx.generation = eng.train_generation()    # identifies this forward among the live ones (include/vsrcap.h, vsr_train_select)
        ctx.token = eng.note_forward()             # while this node is alive and not differentiated, its buffers are not reused
        ctx.eng = eng
        ctx.device = device
        ctx.shapes = [tuple(p.shape) for p in params]
        ctx.save_for_backward(out, gate)           # the library reads them again in the backward pass
        return out, gate

    @staticmethod
    def backward(ctx, g_out, g_gate):
        out, gate = ctx.saved_tensors
        if g_out is None:
            g_out = torch.zeros_like(out)
        if g_gate is None:
            g_gate = torch.zeros_like(gate)
        sink = ctx.eng.grad_sink
        grads = ctx.eng.train_backward(ctx.device, g_out.contiguous(), g_gate.contiguous(), ctx.shapes, ctx.generation, into=sink)
        if sink is not None:
            # training-loop mode (parallel.FlatGrads): the gradients went straight into the caller's flat buffer, whose views
            # ARE the parameters' .grad - nothing for autograd to accumulate (and no 285 MB of fresh tensors per step)
            return (None,) * (5 + len(grads))
        return (None, None, None, None, None) + tuple(grads)


def _params_in_abi_order(model):
    sd = dict(model.named_parameters())
    return [sd[k] for k in _PARAM_KEYS]


def xe_forward_with_grad(model, eng, det, captions, ctrl_seq):
    """forward() under autograd: prepare() has been called with ctrl_seq (one slot per step)."""
    return _DecoderFn.apply(eng, det.size(0), det.device, captions, None, *_params_in_abi_order(model))


def sample_logprobs_with_grad(model, eng, det, ctrl, outs, lps):
    """log-probs of given samples WITH a graph: replays the sampled words / gates through the training forward
    (word fed at step t = previous sample, slot pointer = clamped running sum of the previous gates)."""
    words, gates = outs
    B, T = words.shape
    L = ctrl.size(1)                      # (dense tensor or IndexedRegions: both answer size(1) with the slot count)
    bos = torch.full((B, 1), model.bos_idx, dtype=torch.int64, device=words.device)
    word_in = torch.cat([bos, words[:, :-1]], 1)
    slots = torch.cat([torch.zeros_like(bos), torch.clamp(torch.cumsum(gates[:, :-1], 1), max=L - 1)], 1)
    out, gate = _DecoderFn.apply(eng, B, det.device, word_in, slots, *_params_in_abi_order(model))
    lp_w = out.gather(2, words.unsqueeze(-1)).squeeze(-1)
    lp_g = gate.gather(2, gates.unsqueeze(-1)).squeeze(-1)
    return lp_w, lp_g
