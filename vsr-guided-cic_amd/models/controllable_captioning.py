"""ControllableCaptioningModel on MI355X: the reference's class surface over libvsrcap.so.

Mirrors /root/reference/models/controllable_captioning.py:
  ctor signature and the 28 state_dict keys / shapes        :11-70
  init_weights distributions (xavier-normal, orthogonal W_hh, zero biases)   :72-107
  init_state / step / step_v / test / sample_rl signatures   :109-303
Parameters are ordinary torch Parameters (so .to(), .parameters(), state_dict(), optimizers work); the
library BORROWS their storage, nothing is copied.  All compute is in the HIP library: there is no
PyTorch/CPU fallback, and calling a compute method with CPU tensors raises.
"""
import json
import os

import torch
from torch import nn

from .CaptioningModel import CaptioningModel
from vsrcap.engine import Engine


# GEMM flavour a new model starts in (all three keep fp32 operands in memory and accumulate in fp32):
#   'f16x2' (round 4): every fp32 operand = two fp16 terms under a power-of-two scale, three MFMAs per product, weights pre-split into
#           fp16-pair images per weight version (csrc/gemm_h2.h); the backward pass and sizes that are not multiples of 8 run as 'f32x3'
#   'f32x3' (round 3): three bf16 terms per operand, six MFMAs per product (csrc/gemm_x3.h, gemm_x3s.h); launches of 129 .. 192 rows
#           take the exact kernels (include/vsrcap.h, vsr_set_gemm_mode, has the routing)
#   'f32':  the exact k-ordered fma chain for every launch
# The whole GPU suite runs in every one of them (tests/conftest.py parametrises this default), which is what admits a default.
DEFAULT_COMPUTE_DTYPE = 'f16x2'
COMPUTE_DTYPES = ('f32', 'f32x3', 'f16x2', 'bf16')


# ---- the weights' GENERATION ------------------------------------------------------------------------------------------------
# The reference reads its parameters live at every step (controllable_captioning.py:151-152,177-178; coco_scripts/train.py:112-113 steps
# the optimizer and the next model(...) at :103 sees the new weights).  Here three things are DERIVED from the weights and have to be
# redone when they move: the fp16-pair images (f16x2), the bf16 copies (bf16) and the decode cache (inference).  What says "they moved":
#   (1) model._wgen, bumped by invalidate_cache(), train() / eval(), load_state_dict(), attach_optimizer()'s hook - and by EVERY forward
#       that builds a graph (training: the weights are expected to move between two such calls, whoever moves them);
#   (2) _OPT_STEPS, a process-wide count of torch.optim.Optimizer.step() calls (a global post-step hook): any optimizer stepping
#       anywhere voids every model's derived state, so an evaluation between two steps never sees stale images;
#   (3) sum(p._version) - in-place edits through torch ops.  NOT sufficient on its own: fused optimizers (Adam / SGD(fused=True))
#       update the parameters in place and leave _version untouched on this torch build (round-5 review).
# Writes none of these sees (p.data edits, DLPack / custom kernels into the same storage) in eval mode: call invalidate_cache().
_OPT_STEPS = [0]
_OPT_HOOK = []


def _count_optimizer_steps():
    if _OPT_HOOK:
        return
    try:
        from torch.optim.optimizer import register_optimizer_step_post_hook
    except ImportError:                  # (older torch: training forwards and attach_optimizer() still cover the training loop)
        _OPT_HOOK.append(None)
        return

    def _after_step(opt, args, kwargs):
        _OPT_STEPS[0] += 1
    _OPT_HOOK.append(register_optimizer_step_post_hook(_after_step))


def set_default_compute_dtype(dtype):
    """compute dtype of models constructed from now on ('f32' = exact fma chain everywhere, 'f32x3', 'f16x2', 'bf16'); returns the old one"""
    global DEFAULT_COMPUTE_DTYPE
    if dtype not in COMPUTE_DTYPES:
        raise ValueError("compute dtype must be one of %s" % (COMPUTE_DTYPES,))
    old, DEFAULT_COMPUTE_DTYPE = DEFAULT_COMPUTE_DTYPE, dtype
    return old


class ControllableCaptioningModel(CaptioningModel):
    def __init__(self, seq_len, vocab_size, bos_idx, det_feat_size=2048, input_encoding_size=1000, rnn_size=1000,
                 att_size=512, h2_first_lstm=True, img_second_lstm=False, dataset='coco', *, verb_2_vob_all=None):
        super().__init__(seq_len)
        self.vocab_size = vocab_size
        self.bos_idx = bos_idx
        self.det_feat_size = det_feat_size
        self.input_encoding_size = input_encoding_size
        self.rnn_size = rnn_size
        self.att_size = att_size
        self.h2_first_lstm = h2_first_lstm
        self.img_second_lstm = img_second_lstm

        # verb id -> admissible vocabulary ids, used by step_v (:25-34).  The reference opens the JSON
        # tables relative to the CWD; verb_2_vob_all=... (keyword-only extension) supplies the table directly.
        if verb_2_vob_all is not None:
            self.verb_2_vob_all, self.verb_2_vob = dict(verb_2_vob_all), {}
        else:
            folder, names = (('datasets/coco', ('verb_2_vob_all_refine.json', 'verb_2_vob.json')) if dataset == 'coco'
                             else ('datasets/flickr', ('verb_2_vob_all_refine_flickr.json', 'verb_2_vob_flickr.json')))
            with open(os.path.join(folder, names[0])) as f:
                self.verb_2_vob_all = json.load(f)
            with open(os.path.join(folder, names[1])) as f:
                self.verb_2_vob = json.load(f)

        H, D, E, A = rnn_size, det_feat_size, input_encoding_size, att_size
        in1 = D + E + (H if h2_first_lstm else 0)
        in2 = H + D + (D if img_second_lstm else 0)
        self.embed = nn.Embedding(vocab_size, E)
        self.W1_is = nn.Linear(in1, H)
        self.W1_hs = nn.Linear(H, H)
        self.att_va = nn.Linear(D, A, bias=False)
        self.att_ha = nn.Linear(H, A, bias=False)
        self.att_a = nn.Linear(A, 1, bias=False)
        self.att_sa = nn.Linear(H, A, bias=False)
        self.att_s = nn.Linear(A, 1, bias=False)
        self.lstm_cell_1 = nn.LSTMCell(in1, H)
        self.lstm_cell_2 = nn.LSTMCell(in2, H)
        self.out_fc = nn.Linear(H, vocab_size)
        self.s_fc = nn.Linear(H, D)
        self.W1_ig = nn.Linear(in1, H)
        self.W1_hg = nn.Linear(H, H)
        self.att_ga = nn.Linear(H, A, bias=False)
        self.att_g = nn.Linear(A, 1, bias=False)
        self.init_weights()
        self._eng = None
        self._verb_dev = None
        self._wgen = 0                   # the weights' generation (module comment above)
        _count_optimizer_steps()
        # prepare() caches the hoisted per-image tensors keyed on (data_ptr, tensor._version, shapes, weights generation);
        # set force_prepare = True when inputs / weights are rewritten in ways that do not bump _version
        # (t.data.copy_, DLPack / custom-kernel writes, p.data edits), or call invalidate_cache() after such a write
        self.force_prepare = False
        # set_valid_rows_bound(): the caller's upper bound on the non-padding region rows of the NEXT calls (None: the library counts them
        # and reads the count back, its one host synchronisation per call)
        self.valid_rows_bound = None
        # VSR_COMPUTE_DTYPE lets a whole test run (or an unchanged reference script) select the GEMM flavour without code changes
        self.compute_dtype = os.environ.get('VSR_COMPUTE_DTYPE', DEFAULT_COMPUTE_DTYPE)

    def set_compute_dtype(self, dtype):
        """'f16x2' (default) / 'f32x3': fp32 operands and fp32 accumulation, products formed on the fp16 / bf16 matrix cores from two /
        three terms per operand (three / six MFMAs per product; csrc/gemm_h2.h / gemm_x3.h; the module comment above and
        include/vsrcap.h have the routing).  Token parity and the 1e-4 loss bound hold (every GPU test runs in each flavour).
        'f32': the exact k-ordered fp32 fma chain (v_mfma_f32_32x32x2_f32) for every launch.
        'bf16': throughput mode - matrix products take bf16 operands with fp32 accumulation (v_mfma_f32_32x32x16_bf16);
        parameters, optimizer state, states and reductions stay fp32.  Not a parity mode."""
        if dtype not in COMPUTE_DTYPES:
            raise ValueError("compute dtype must be one of %s" % (COMPUTE_DTYPES,))
        self.compute_dtype = dtype
        return self

    def set_valid_rows_bound(self, n):
        """Extension (not in the reference): an upper bound on the rows of the region tensor that are not zero padding - for index lists,
        on the non-zero rows of the feature bank - which a caller that builds its inputs on the host (coco_scripts/eval_coco.py:222-237,
        data/field.py:44-61) has for free.  The decode / training calls then never wait for the device (include/vsrcap.h,
        vsr_set_valid_rows_bound).  None switches back to the device-side count + read-back.  A bound that is too small cannot fail the call
        (nothing is read back): the rows beyond it get a ZERO att_va projection and are counted - the input-contract check
        (VSR_CHECK_IDS=1 / engine.check_ids) turns that count into an IndexError; use it when the bound is not known to be safe."""
        self.valid_rows_bound = None if n is None else int(n)
        return self

    def invalidate_cache(self):
        """the weights (or inputs) were rewritten in a way nothing above can see: everything derived from them is redone by the next call"""
        self._wgen += 1
        if self._eng is not None:
            self._eng.invalidate()

    def attach_optimizer(self, optimizer):
        """Extension: tie an optimizer's step() to this model's weight generation (a per-optimizer post-step hook).  torch optimizers are
        already counted process-wide (_OPT_STEPS); this is for optimizer classes that do not run torch's step hooks.  Returns the hook
        handle (or None when the object has no register_step_post_hook)."""
        reg = getattr(optimizer, "register_step_post_hook", None)
        if reg is None:
            return None
        return reg(lambda opt, args, kwargs: self.invalidate_cache())

    def train(self, mode=True):
        # a mode switch redoes the derived state: eval() after a training run must decode with the FINAL weights whatever moved them
        if mode != self.training:
            self._wgen = getattr(self, "_wgen", 0) + 1
        return super().train(mode)

    def load_state_dict(self, *a, **kw):
        self._wgen = getattr(self, "_wgen", 0) + 1
        return super().load_state_dict(*a, **kw)

    def init_weights(self):
        for name, p in self.named_parameters():
            if name.endswith('weight_hh'):
                nn.init.orthogonal_(p)
            elif p.dim() == 2:
                nn.init.xavier_normal_(p)
            else:
                nn.init.constant_(p, 0)

    def init_state(self, b_s, device):
        z = lambda: torch.zeros((b_s, self.rnn_size), dtype=torch.float32, device=device)
        return (z(), z()), (z(), z()), torch.zeros((b_s,), dtype=torch.long, device=device)

    # ------------------------------------------------------------------ engine plumbing
    def _engine(self, device, weights_may_have_moved=False):
        """weights_may_have_moved: the caller builds an autograd graph (a training forward): the derived weight state is redone"""
        if weights_may_have_moved:
            self._wgen += 1
        if device.type != 'cuda':
            raise RuntimeError("ControllableCaptioningModel (MI355X build) computes only on the GPU: move the model and "
                               "its inputs to 'cuda'. There is no CPU fallback.")
        if device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        if self._eng is None or self._eng.device != device:
            if self._eng is not None and self._eng.grad_sink is not None:
                # a DataParallelStep holds this engine and its flat gradient buffer: a silent replacement would send the next
                # backward to autograd's in-place accumulation into views nobody zeroes
                raise RuntimeError("the model moved from %s to %s while a gradient sink (parallel.DataParallelStep) is attached to "
                                   "its engine: build the DataParallelStep after the move" % (self._eng.device, device))
            # one handle per device; its launches run under torch.cuda.device(device) whatever the caller's current device is
            self._eng = Engine(dict(seq_len=self.seq_len, vocab_size=self.vocab_size, bos_idx=self.bos_idx,
                                    det_feat_size=self.det_feat_size, input_encoding_size=self.input_encoding_size,
                                    rnn_size=self.rnn_size, att_size=self.att_size,
                                    h2_first_lstm=int(self.h2_first_lstm), img_second_lstm=int(self.img_second_lstm)), device)
            self._verb_dev = None
        if self.force_prepare:
            self._eng.invalidate()
        params = {k: v.data for k, v in self.named_parameters()}
        pdev = next(iter(params.values())).device
        if pdev.type != 'cuda' or (device.index is not None and pdev.index != device.index):
            raise RuntimeError("model parameters are on %s but the inputs are on %s" % (pdev, device))
        self._eng.bind(params)
        self._eng.set_bf16(pdev, self._weights_version(), self.compute_dtype == 'bf16')
        self._eng.set_gemm_mode(self.compute_dtype in ('f32x3', 'f16x2'))
        dims8 = all(v % 8 == 0 for v in (self.det_feat_size, self.input_encoding_size, self.rnn_size, self.att_size))
        self._eng.set_h2(pdev, self._weights_version(), self.compute_dtype == 'f16x2' and dims8)      # (other sizes: plain f32x3)
        # inference keeps a weight-only cache (embedding projection); while training the weights move every step
        self._eng.decode_cache(pdev, self._weights_version(), enable=not self.training)
        return self._eng

    def _weights_version(self):
        return (self._wgen, _OPT_STEPS[0], sum(p._version for p in self.parameters()))

    def _builds_graph(self):
        return torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())

    def _verbs(self, eng, verbs, device, rows=None, L=None):
        if rows is not None and tuple(verbs.shape) != (rows, L):
            raise RuntimeError("verb list statics[2] must have shape (rows, slots) = (%d, %d), got %s" % (rows, L, tuple(verbs.shape)))
        if self._verb_dev != device:
            eng.set_verb_table(self.verb_2_vob_all, device)
            self._verb_dev = device
        return verbs.to(device=device, dtype=torch.float32).contiguous()

    @staticmethod
    def _n_slots(regions):
        from vsrcap.regions import IndexedRegions
        return regions.slot_idx.size(1) if isinstance(regions, IndexedRegions) else regions.size(1)

    def _prepare(self, eng, det, regions, beam, for_training=False):
        """Hoisted per-image work for either region format: the reference's dense (B,L,R,D) tensor, or
        vsrcap.regions.IndexedRegions (index lists into the image's feature bank; training too when every row is its own image)."""
        from vsrcap.regions import IndexedRegions
        if isinstance(regions, IndexedRegions):
            return eng.prepare_indexed(det, regions.bank, regions.slot_idx, regions.row_img, beam, self._weights_version(), self.valid_rows_bound,
                                       for_training=for_training)
        return eng.prepare(det, regions, beam, self._weights_version(), self.valid_rows_bound, for_training=for_training)

    # ------------------------------------------------------------------ loops (CaptioningModel hooks)
    def _run_forward(self, statics, seqs):
        det, (captions, ctrl_seq) = statics[0], seqs
        with_grad = self._builds_graph()
        eng = self._engine(det.device, weights_may_have_moved=with_grad)
        if captions.size(1) > self.seq_len:
            raise RuntimeError("captions longer than seq_len")
        from vsrcap.regions import IndexedRegions
        if with_grad:
            if isinstance(ctrl_seq, IndexedRegions) and ctrl_seq.row_img is not None:
                raise RuntimeError("training on IndexedRegions needs one decoder row per image (row_img=None), as the reference's "
                                   "training batches have; with a row -> image map train on regions.dense()")
            B = self._prepare(eng, det, ctrl_seq, 1, for_training=True)
            from vsrcap.train import xe_forward_with_grad
            return xe_forward_with_grad(self, eng, det, captions, ctrl_seq)
        B = self._prepare(eng, det, ctrl_seq, 1)
        return eng.xe_forward(B, det.device, captions)

    def _run_greedy(self, statics):
        """test() of the reference takes (detections, ctrl_det_seqs_test) only (:299); a third static (verb list) is an
        extension that runs the greedy loop over step_v (gt=False) == beam_search_v with beam_size 1."""
        det, ctrl = statics[0], statics[1]
        if det.shape[0] == 0:        # the reference's test() unrolls over an empty batch and returns (0, T) outputs (CaptioningModel.py:38-52)
            e = torch.zeros(0, self.seq_len, dtype=torch.int64, device=det.device)
            return e, e.clone()
        eng = self._engine(det.device)
        B = self._prepare(eng, det, ctrl, 1)
        v = self._verbs(eng, statics[2], det.device, B, self._n_slots(ctrl)) if len(statics) > 2 and statics[2] is not None else None
        return eng.greedy(B, det.device, v, False)

    def _run_sample(self, statics, seed=None, forced=None):
        det, ctrl = statics[0], statics[1]
        with_grad = self._builds_graph()
        eng = self._engine(det.device, weights_may_have_moved=with_grad)
        from vsrcap.regions import IndexedRegions
        if with_grad and isinstance(ctrl, IndexedRegions) and ctrl.row_img is not None:
            raise RuntimeError("sample_rl with gradients on IndexedRegions needs one decoder row per image (row_img=None)")
        B = self._prepare(eng, det, ctrl, 1, for_training=with_grad)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            # data-parallel ranks that called torch.manual_seed(s) with the same s would otherwise draw IDENTICAL Gumbel /
            # uniform noise for their shard rows (the Philox stream is keyed by (seed, local row, t))
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                seed ^= (dist.get_rank() + 1) << 48
        outs, lps = eng.sample(B, det.device, seed, forced)
        if with_grad:
            from vsrcap.train import sample_logprobs_with_grad
            lps = sample_logprobs_with_grad(self, eng, det, ctrl, outs, lps)
        return outs, lps

    def _run_beam(self, statics, eos_idxs, beam_size, out_size, with_verbs, gt):
        det, ctrl = statics[0], statics[1]
        eng = self._engine(det.device)
        B = self._prepare(eng, det, ctrl, beam_size)
        v = self._verbs(eng, statics[2], det.device, B, self._n_slots(ctrl)) if with_verbs else None
        (w, g), (lw, lg), _ = eng.beam(B, det.device, beam_size, out_size, eos_idxs[0], eos_idxs[1], v, gt)
        if out_size == 1:
            return [w[:, 0], g[:, 0]], [lw[:, 0], lg[:, 0]]
        return [w, g], [lw, lg]

    # ------------------------------------------------------------------ single timestep (feedback mode)
    def step(self, t, state, prev_outputs, statics, seqs, *args, mode='teacher_forcing'):
        return self._step(t, state, prev_outputs, statics, seqs, mode, False, False)

    def step_v(self, t, state, prev_outputs, statics, seqs, *args, mode='teacher_forcing', gt=False):
        return self._step(t, state, prev_outputs, statics, seqs, mode, True, gt)

    def _step(self, t, state, prev_outputs, statics, seqs, mode, with_verbs, gt):
        assert (mode in ['teacher_forcing', 'feedback'])
        det = statics[0]
        eng = self._engine(det.device)
        if mode == 'teacher_forcing':
            # one slot per row: regions of step t, word of step t (controllable_captioning.py:131-133)
            regions = seqs[1][:, t:t + 1].contiguous()
            eng.prepare(det, regions, 1, self._weights_version())
            (h1, c1), (h2, c2), slot = state
            prev = (seqs[0][:, t], torch.zeros_like(slot))
            outs, (s1, s2, _) = eng.step(1, 1, prev, ((h1, c1), (h2, c2), torch.zeros_like(slot)))
            # feeding the ground-truth word as "previous output" with gate 0 on a single slot IS teacher forcing
            return outs, (s1, s2, slot)
        B = self._prepare(eng, det, statics[1], 1)
        v = self._verbs(eng, statics[2], det.device, B, self._n_slots(statics[1])) if with_verbs else None
        return eng.step(t, 1, prev_outputs, state, v, gt)

    def test(self, detections, ctrl_det_seqs_test):
        return super().test((detections, ctrl_det_seqs_test))

    def sample_rl(self, detections, ctrl_det_seqs_test, **kw):
        return super().sample_rl((detections, ctrl_det_seqs_test), **kw)
