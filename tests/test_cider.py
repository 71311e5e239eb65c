"""SCST reward: the CIDEr-D restatement (oracle, CPU) and the device kernel against it (GPU).
Parity unpinned: speaksee is absent, so these tests pin the oracle to the algorithm's defining properties and the kernel to
the oracle."""
import math

import numpy as np
import pytest

import cider_oracle as co


def _corpus(rng, n_samples, V, max_refs=3):
    out = []
    for _ in range(n_samples):
        refs = []
        for _ in range(rng.randint(1, max_refs + 1)):
            refs.append([int(x) for x in rng.randint(4, V, size=rng.randint(3, 15))])
        out.append(refs)
    return out


def test_oracle_properties():
    rng = np.random.RandomState(0)
    corpus = _corpus(rng, 50, 30)
    c = co.CiderD(corpus)
    ref = [5, 6, 7, 8, 9, 10]
    assert abs(c.score([ref], ref) - 10.0) < 1e-9                 # identical caption: cosine 1 at every order, no length penalty
    assert c.score([ref], []) == 0.0 and c.score([ref], [29, 28, 27, 26]) == 0.0      # nothing in common
    short = [5, 6, 7]                                              # 3 tokens: no 4-grams -> that order contributes 0
    assert abs(c.score([short], short) - 7.5) < 1e-9
    # the length penalty acts on bigram counts: a prefix of the reference scores below exp(-(delta^2) / 72) * 10
    s = c.score([ref], ref[:4])
    assert 0 < s < 10.0 * math.exp(-(2 ** 2) / 72.0) + 1e-9
    # an n-gram every sample contains has idf 0 and adds nothing
    every = co.CiderD([[[1, 2, 3]], [[1, 2, 4]], [[1, 2, 5]]])
    assert every.score([[1, 2, 9]], [1, 2, 8]) == 0.0
    # more references: the mean over references
    a, b = c.score([ref], ref), c.score([[11, 12, 13, 14]], ref)
    assert abs(c.score([ref, [11, 12, 13, 14]], ref) - (a + b) / 2) < 1e-9


def test_clean_matches_the_callers_text_pipeline():
    # decode-to-eos, groupby de-duplication BEFORE punctuation is dropped (train.py:154, :161, :165-167)
    assert co.clean([7, 7, 8, 2, 8, 8, 3, 9], eos=3, drop={2}) == [7, 8, 8]
    assert co.clean([3, 5], eos=3) == []
    from vsrcap import reward
    assert reward.clean_ids([7, 7, 8, 2, 8, 8, 3, 9], eos=3, drop={2}) == [7, 8, 8]


@pytest.mark.gpu
def test_device_cider_matches_oracle():
    import torch
    from vsrcap import reward
    rng = np.random.RandomState(1)
    V, eos, pad = 60, 3, 0
    corpus = _corpus(rng, 200, V)
    c = co.CiderD(corpus)
    dev = reward.CiderD(corpus, V)
    N, T, n_ref, Tr = 64, 20, 3, 18
    drop = np.zeros(V, dtype=np.uint8)
    drop[[4, 5]] = 1
    cand = rng.randint(4, V, size=(N, T)).astype(np.int64)
    refs = np.full((N, n_ref, Tr), pad, dtype=np.int64)
    for i in range(N):
        src = corpus[i % len(corpus)]
        for r in range(n_ref):
            w = src[r % len(src)]
            refs[i, r, :len(w)] = w
            if len(w) < Tr:
                refs[i, r, len(w)] = eos
        if i % 4 == 0:                                   # a candidate that copies most of a reference
            w = src[0]
            cand[i, :len(w)] = w
            cand[i, len(w):] = eos
        if i % 4 == 1:
            cand[i, rng.randint(2, T)] = eos             # early end
        if i % 4 == 2:
            cand[i, 3:6] = cand[i, 2]                    # a run of repeats
    got = dev.rewards(torch.from_numpy(cand).cuda(), torch.from_numpy(refs).cuda(), eos, pad, torch.from_numpy(drop).cuda()).cpu().numpy()
    dropset = {4, 5}
    want = np.array([c.score([co.clean(refs[i, r], eos=eos, drop=dropset) if refs[i, r, 0] != pad else [] for r in range(n_ref)],
                             co.clean([w for w in cand[i] if w != pad], eos=eos, drop=dropset)) for i in range(N)])
    assert got.dtype == np.float32 and (want > 0).sum() > N // 4
    np.testing.assert_allclose(got, want.astype(np.float32), rtol=1e-5, atol=1e-6)
    with pytest.raises(RuntimeError):
        dev.rewards(torch.from_numpy(cand), torch.from_numpy(refs).cuda(), eos)          # CPU tensor: no fallback
