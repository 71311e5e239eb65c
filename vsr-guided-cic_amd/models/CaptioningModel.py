"""Decode / train loop surface of the reference's CaptioningModel (models/CaptioningModel.py:8-294).

The reference unrolls `self.step` in Python, once per timestep and once per ATen op; here every loop is
ONE call into libvsrcap.so, which runs all T steps on the device (hand-written HIP, no host sync).  The
method names, argument meaning and returned tuple shapes are the reference's:

  forward(statics, seqs)                         :22   -> (logp_words (B,T,V), logp_gates (B,T,2))
  test(statics)                                  :38   -> (words (B,T), gates (B,T)) int64
  sample_rl(statics)                             :54   -> ((words, gates), (lp_words, lp_gates))
  beam_search(statics, eos_idxs, beam_size, out_size=1)            :116
  beam_search_v(statics, eos_idxs, beam_size, out_size=1, gt=False) :197
       -> ([words, gates], [lp_words, lp_gates]); (B,T) when out_size == 1 else (B,out_size,T)
"""
from torch import nn


class CaptioningModel(nn.Module):
    def __init__(self, seq_len):
        self.seq_len = seq_len
        super().__init__()

    # subclass contract (same three hooks as the reference :13-20)
    def init_weights(self):
        raise NotImplementedError

    def init_state(self, b_s, device):
        raise NotImplementedError

    def step(self, t, state, prev_outputs, images, seqs, *args, mode='teacher_forcing'):
        raise NotImplementedError

    # device loops, implemented by the subclass on top of the C ABI
    def _run_forward(self, statics, seqs):
        raise NotImplementedError

    def _run_greedy(self, statics):
        raise NotImplementedError

    def _run_sample(self, statics, **kw):
        raise NotImplementedError

    def _run_beam(self, statics, eos_idxs, beam_size, out_size, with_verbs, gt):
        raise NotImplementedError

    def forward(self, statics, seqs, *args):
        return self._run_forward(statics, seqs)

    def test(self, statics, *args):
        return self._run_greedy(statics)

    def sample_rl(self, statics, *args, **kw):
        return self._run_sample(statics, **kw)

    def beam_search(self, statics, eos_idxs, beam_size, out_size=1, *args):
        return self._run_beam(statics, eos_idxs, beam_size, out_size, False, False)

    def beam_search_v(self, statics, eos_idxs, beam_size, out_size=1, *args, gt=False):
        return self._run_beam(statics, eos_idxs, beam_size, out_size, True, gt)
