"""Shared builders for the parity tests: regenerate the closed-form weights / inputs of a fixture."""
import numpy as np
import torch

from vsrcap import synth


def weights_for(cfg, gains=None, wseed=0, **kw):
    return synth.make_weights(cfg["V"], cfg["D"], cfg["E"], cfg["H"], cfg["A"], seed=wseed, gains=gains, **kw)


def decode_inputs(cfg, seed, n=None):
    det = synth.make_detections(cfg["B"], cfg["R0"], cfg["D"], seed=seed)
    ctrl = synth.make_ctrl(cfg["B"], cfg["L"], cfg["R"], cfg["D"], seed=seed)
    if n is not None:
        det, ctrl = det[:n], ctrl[:n]
    return torch.from_numpy(det), torch.from_numpy(ctrl)


def train_inputs(cfg, seed):
    det = torch.from_numpy(synth.make_detections(cfg["B"], cfg["R0"], cfg["D"], seed=seed))
    ctrl_seq = torch.from_numpy(synth.make_ctrl(cfg["B"], cfg["T"], cfg["R"], cfg["D"], seed=seed + 1000))
    caps = torch.from_numpy(synth.make_captions(cfg["B"], cfg["T"], cfg["V"], seed=seed))
    gts = torch.from_numpy(synth.make_gate_gts(cfg["B"], cfg["T"], seed=seed))
    return det, ctrl_seq, caps, gts


def build_model(cfg, weights, device, bos=2, verb_table=None, **kw):
    from models import ControllableCaptioningModel
    m = ControllableCaptioningModel(cfg["T"], cfg["V"], bos, det_feat_size=cfg["D"], input_encoding_size=cfg["E"],
                                    rnn_size=cfg["H"], att_size=cfg["A"], verb_2_vob_all=verb_table or {}, **kw)
    m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in weights.items()})
    return m.to(device).eval()
