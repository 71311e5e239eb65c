"""SCST reward on the GPU: per-sample CIDEr-D on token ids (`include/vsrcap.h`: vsr_cider_rewards).

Replaces the host section of the reference's RL step (`coco_scripts/train.py:154-172`): the sampled ids never leave the
device.  `CiderD(corpus_refs, vocab_size)` plays the role of `evaluation.Cider(PTBTokenizer.tokenize(ref_caps_train))`
(train.py:67): it builds the corpus document-frequency table once; `rewards()` is `cider_train.compute_score(...)[1]`.
The text clean-up of the caller happens on ids: stop at <eos>, collapse consecutive repeats (`itertools.groupby`,
train.py:161), drop the ids of punctuation tokens (what `PTBTokenizer` removes).  Parity unpinned: speaksee is not in
this image; the algorithm is the published CIDEr-D, restated in `oracle/cider_oracle.py`.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib


def _ngram_keys(words, k):
    """uint64 keys of the order-k n-grams of one cleaned caption (ids packed 16 bits each, first id lowest)."""
    w = np.asarray(words, dtype=np.uint64)
    n = len(w) - k + 1
    if n <= 0:
        return np.zeros(0, dtype=np.uint64)
    key = np.zeros(n, dtype=np.uint64)
    for j in range(k):
        key |= w[j:j + n] << np.uint64(16 * j)
    return key


def clean_ids(ids, eos=None, pad=None, drop=()):
    out = []
    for w in ids:
        w = int(w)
        if w == eos or w == pad:
            break
        if out and w == out[-1][0]:
            continue
        out.append((w, w in drop))
    return [w for w, d in out if not d]


class CiderD:
    def __init__(self, corpus_refs, vocab_size, n=4, sigma=6.0):
        """corpus_refs: one entry per training sample, each a list of reference token-id lists (already cleaned)."""
        if n != 4:
            raise ValueError("the device kernel is written for n = 4")
        if vocab_size > 65535:
            raise ValueError("vocabulary of %d ids does not fit the 16-bit n-gram key" % vocab_size)
        self.vocab_size, self.sigma = int(vocab_size), float(sigma)
        self.ref_len = math.log(float(len(corpus_refs)))
        self.keys, self.idf = [], []
        for k in range(1, 5):
            per_sample = [np.unique(np.concatenate([_ngram_keys(r, k) for r in refs] + [np.zeros(0, dtype=np.uint64)])) for refs in corpus_refs]
            allk = np.concatenate(per_sample) if per_sample else np.zeros(0, dtype=np.uint64)
            keys, df = np.unique(allk, return_counts=True)
            self.keys.append(keys.astype(np.uint64))
            self.idf.append(self.ref_len - np.log(np.maximum(1.0, df.astype(np.float64))))
        self._dev = None

    def to(self, device):
        self._dev = (device,
                     [torch.from_numpy(k.view(np.int64).copy()).to(device) if len(k) else torch.zeros(1, dtype=torch.int64, device=device) for k in self.keys],
                     [torch.from_numpy(v.copy()).to(device) if len(v) else torch.zeros(1, dtype=torch.float64, device=device) for v in self.idf])
        return self

    def rewards(self, cand, refs, eos, pad=-1, drop_mask=None):
        """cand (N, T) int64 GPU; refs (N, n_ref, Tr) int64 GPU (padded with `pad` or ended by `eos`);
        drop_mask (V,) uint8 GPU or None.  Returns (N,) fp32 rewards on the GPU, nothing is copied to the host."""
        if self._dev is None or self._dev[0] != cand.device:
            self.to(cand.device)
        _, keys, idf = self._dev
        if cand.dtype != torch.int64 or refs.dtype != torch.int64 or not cand.is_cuda or not refs.is_cuda:
            raise RuntimeError("candidates and references must be int64 GPU tensors")
        cand, refs = cand.contiguous(), refs.contiguous()
        N, T = cand.shape
        if refs.dim() != 3 or refs.size(0) != N:
            raise RuntimeError("refs must be (N, n_ref, Tr)")
        if drop_mask is not None and (drop_mask.dtype != torch.uint8 or drop_mask.numel() != self.vocab_size or not drop_mask.is_cuda):
            raise RuntimeError("drop_mask must be a (V,) uint8 GPU tensor")
        out = torch.empty(N, dtype=torch.float32, device=cand.device)
        lib = _lib.load()
        kp = (C.c_void_p * 4)(*[k.data_ptr() for k in keys])
        ip = (C.c_void_p * 4)(*[v.data_ptr() for v in idf])
        cnt = (C.c_int32 * 4)(*[len(k) for k in self.keys])
        stream = C.c_void_p(torch.cuda.current_stream(cand.device).cuda_stream)
        _lib.check(lib.vsr_cider_rewards(kp, ip, cnt, C.c_double(self.ref_len), C.c_void_p(cand.data_ptr()), N, T,
                                         C.c_void_p(refs.data_ptr()), refs.size(1), refs.size(2), int(eos), int(pad),
                                         C.c_void_p(drop_mask.data_ptr() if drop_mask is not None else 0), self.vocab_size,
                                         C.c_double(self.sigma), C.c_void_p(out.data_ptr()), stream))
        return out
