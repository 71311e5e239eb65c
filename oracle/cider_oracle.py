"""CPU oracle for the SCST reward.  TEST INFRASTRUCTURE ONLY: imported by tests/ (and by bench.py's
CPU-baseline leg), never by the product path.

PARITY UNPINNED.  The reference takes its reward from `speaksee.evaluation.Cider` (speaksee==0.0.1, `vsr.yml:247`,
un-vendored and absent from this image; call sites `coco_scripts/train.py:67, 154-172`).  This file restates the PUBLISHED
algorithm that class wraps -- CIDEr-D (Vedantam et al., CVPR 2015) as implemented by coco-caption's `cider_scorer.py`,
which speaksee's scorer follows: n = 1..4, tf-idf vectors with corpus document frequencies, clipped cosine similarity,
Gaussian length penalty (sigma = 6), mean over n, x10 -- on TOKEN-ID sequences, together with the id-level equivalent of
the caller's text clean-up:
  * `TextField.decode`: the caption ends at the first <eos>                     (train.py:154)
  * `' '.join(k for k, g in itertools.groupby(gen_i))`: consecutive repeats collapse (train.py:161)
  * `PTBTokenizer.tokenize`: punctuation tokens are dropped (a per-vocabulary drop set; words are already lower-case
    vocabulary entries)                                                          (train.py:165-167)
No golden vector of the reference exists for this path (no test, no fixture, package absent); the GPU kernel is checked
against this restatement only.
"""
import math
from collections import defaultdict


def clean(ids, eos=None, drop=()):
    """decode-to-eos, collapse consecutive repeats, drop punctuation ids"""
    out = []
    for w in ids:
        w = int(w)
        if eos is not None and w == eos:
            break
        out.append(w)
    dedup = [w for i, w in enumerate(out) if i == 0 or w != out[i - 1]]
    return [w for w in dedup if w not in drop]


def cook(words, n=4):
    counts = defaultdict(int)
    for k in range(1, n + 1):
        for i in range(len(words) - k + 1):
            counts[tuple(words[i:i + k])] += 1
    return counts


class CiderD:
    def __init__(self, corpus_refs, n=4, sigma=6.0):
        """corpus_refs: list (one entry per training sample) of lists of reference token lists (already cleaned)."""
        self.n, self.sigma = n, sigma
        self.doc_frequency = defaultdict(float)
        for refs in corpus_refs:
            for ngram in set(ng for ref in refs for ng in cook(ref, n)):
                self.doc_frequency[ngram] += 1
        self.ref_len = math.log(float(len(corpus_refs)))

    def _vec(self, counts):
        vec = [defaultdict(float) for _ in range(self.n)]
        norm = [0.0] * self.n
        length = 0
        for ngram, tf in counts.items():
            df = math.log(max(1.0, self.doc_frequency.get(ngram, 0.0)))
            k = len(ngram) - 1
            vec[k][ngram] = float(tf) * (self.ref_len - df)
            norm[k] += vec[k][ngram] ** 2
            if k == 1:
                length += tf                     # the scorer measures length in bigrams
        return vec, [math.sqrt(x) for x in norm], length

    def score(self, refs, hyp):
        """refs: list of reference token lists; hyp: token list.  Returns the per-sample CIDEr-D (x10)."""
        vh, nh, lh = self._vec(cook(hyp, self.n))
        total = [0.0] * self.n
        for ref in refs:
            vr, nr, lr = self._vec(cook(ref, self.n))
            delta = float(lh - lr)
            for k in range(self.n):
                val = 0.0
                for ngram, x in vh[k].items():
                    val += min(x, vr[k].get(ngram, 0.0)) * vr[k].get(ngram, 0.0)
                if nh[k] != 0 and nr[k] != 0:
                    val /= nh[k] * nr[k]
                val *= math.e ** (-(delta ** 2) / (2 * self.sigma ** 2))
                total[k] += val
        return 10.0 * sum(total) / self.n / len(refs)
