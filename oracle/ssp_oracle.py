"""CPU oracle for the two ordering models that run BEFORE the decoder in the eval loop.  TEST INFRASTRUCTURE ONLY
(same rules as vsr_oracle.py: only tests/, smoke() and bench.py's cpu_baseline leg may import it).

Restated from the equations, not the text, of
  /root/reference/models/sort_model.py:105-183        S_SSP.generate(mode='not-normal'): greedy "pick from the remaining roles"
  /root/reference/models/sort_modules.py:25-135       encoder / decoder stacks (pre-LN; the decoder's cross attention re-uses
                                                      the SELF-attention projections: sort_modules.py:88 calls self.attention)
  /root/reference/models/transformer_modules.py:18-147, 182-215, 302-345   attention (-1e3 mask fill, 1/sqrt(64) scaling),
                                                      embeddings scaled by sqrt(512), feed-forward
  /root/reference/models/sinkhorn_network.py:30-51    SinkhornNet.forward + 20 Sinkhorn iterations (eps "10e-8" = 1e-7)
  /root/reference/coco_scripts/eval_coco.py:183-200   transpose, munkres on max - value, argsort of the assigned columns
Parity PINNED for the two networks: tests/golden/make_golden_ssp.py runs the reference's S_SSP and SinkhornNet (they import
with torch only) on closed-form weights and commits their outputs (g11_ssp.npz).  The assignment step is parity UNPINNED:
munkres (eval_coco.py:13) is not in the image; the oracle computes the optimum of the same cost matrix (exhaustive search up
to 7 x 7 in the generator, scipy.optimize.linear_sum_assignment beyond) - any optimal solver agrees unless two optima tie.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

HID, HEADS, FF, MAXLEN, NROLE = 512, 8, 2048, 10, 26


def _t(sd, k):
    v = sd[k]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))


class SSPOracle:
    def __init__(self, sd, dtype=torch.float32):
        self.p = {k: _t(sd, k).to(dtype) for k in sd if not k.endswith(".pe") and not k.endswith("one_hot")}
        self.dtype = dtype

    def _lin(self, x, name):
        return F.linear(x, self.p[name + ".weight"], self.p[name + ".bias"])

    def _ln(self, x, name):
        return F.layer_norm(x, (HID,), self.p[name + ".weight"], self.p[name + ".bias"], 1e-5)

    def _mha(self, pre, q_in, kv_in, allowed):
        """multi-head attention with the projections `pre`.linear_{Q,K,V,O}; allowed (S,Tq,Tk) bool or None"""
        S, Tq, _ = q_in.shape
        Tk = kv_in.shape[1]
        hd = HID // HEADS
        q = self._lin(q_in, pre + ".linear_Q").view(S, Tq, HEADS, hd).transpose(1, 2)
        k = self._lin(kv_in, pre + ".linear_K").view(S, Tk, HEADS, hd).transpose(1, 2)
        v = self._lin(kv_in, pre + ".linear_V").view(S, Tk, HEADS, hd).transpose(1, 2)
        logits = q @ k.transpose(-2, -1) / math.sqrt(hd)
        if allowed is not None:
            logits = logits.masked_fill(~allowed.unsqueeze(1), -1e3)
        ctx = (F.softmax(logits, -1) @ v).transpose(1, 2).reshape(S, Tq, HID)
        return self._lin(ctx, pre + ".linear_O")

    def encode(self, verbs, roles):
        """verbs (S,) int, roles (S,10) int -> prior states (S,10,512)"""
        p = self.p
        sc = math.sqrt(HID)
        x = p["v_embed_layer.weight"][(verbs % 10000).long()].unsqueeze(1) * sc + p["sr_embed_layer.weight"][roles.long()] * sc
        x = self._lin(x, "encoder.fc_feat")
        for l in range(3):
            pre = "encoder.encoder_layers.%d" % l
            y = self._ln(x, pre + ".layer_norm1")
            x1 = self._mha(pre + ".attention", y, y, None) + x
            y = self._ln(x1, pre + ".layer_norm2")
            x = self._lin(F.relu(self._lin(y, pre + ".ff_layer.w_1")), pre + ".ff_layer.w_2") + x1
        return self._ln(x, "encoder.layer_norm")

    def decode_states(self, tokens, prior):
        """tokens (S,Tq) int -> decoder states (S,Tq,512)"""
        S, Tq = tokens.shape
        x = self.p["sr_embed_layer.weight"][tokens.long()] * math.sqrt(HID)
        causal = torch.tril(torch.ones(Tq, Tq, dtype=torch.bool))
        allowed = causal.unsqueeze(0) & (tokens != 0).unsqueeze(1)            # key j visible to query i: j <= i and token j != 0
        for l in range(3):
            pre = "decoder.encoder_layers.%d" % l
            h = self._ln(x, pre + ".layer_norm1")
            h1 = self._mha(pre + ".attention", h, h, allowed) + x
            h = self._ln(h1, pre + ".layer_norm2")
            h2 = self._mha(pre + ".attention", h, prior, None) + h1            # the SAME projections (sort_modules.py:88)
            h = self._ln(h2, pre + ".layer_norm3")
            x = self._lin(F.relu(self._lin(h, pre + ".ff_layer.w_1")), pre + ".ff_layer.w_2") + h2
        return self._ln(x, "decoder.layer_norm")

    def generate(self, verbs, roles, return_margin=False):
        """S_SSP.generate(mode='not-normal') for S sequences at once (the reference runs batch size 1).
        verbs (S,), roles (S,10) with 0 = padding -> pred (S,10) int64 (0 past the sequence's roles), logp (S,10)
        [, margin (S,): smallest best-minus-runner-up log-prob over the sequence's picks, inf when never contested]."""
        verbs, roles = torch.as_tensor(verbs), torch.as_tensor(roles)
        S = roles.shape[0]
        prior = self.encode(verbs, roles)
        remain = roles != 0
        pred = torch.zeros(S, MAXLEN, dtype=torch.int64)
        logp = torch.zeros(S, MAXLEN, dtype=self.dtype)
        tokens = torch.zeros(S, 1, dtype=torch.int64)
        margin = torch.full((S,), float("inf"))
        for t in range(MAXLEN):
            if not remain.any():
                break
            st = self.decode_states(tokens, prior)[:, -1]
            lp = F.log_softmax(self._lin(st, "expander_nn"), -1)
            nxt = torch.zeros(S, dtype=torch.int64)
            for s in range(S):
                if not remain[s].any():
                    continue
                idx = torch.nonzero(remain[s]).flatten()
                cand = lp[s, roles[s, idx].long()]
                j = int(torch.argmax(cand))                                   # first maximum (torch.max)
                if cand.numel() > 1:
                    top2 = torch.topk(cand, 2)[0]
                    margin[s] = min(float(margin[s]), float(top2[0] - top2[1]))
                pred[s, t] = roles[s, idx[j]]
                logp[s, t] = cand[j]
                remain[s, idx[j]] = False
                nxt[s] = roles[s, idx[j]]
            tokens = torch.cat([tokens, nxt.unsqueeze(1)], 1)
        if return_margin:
            return pred, logp, margin
        return pred, logp


class SinkhornOracle:
    def __init__(self, sd, n_iters=20, tau=0.1, dtype=torch.float32):
        self.p = {k: _t(sd, k).to(dtype) for k in sd}
        self.n_iters, self.tau = n_iters, tau

    def forward(self, seq):
        """seq (Q,N,2352) -> doubly-normalised (Q,N,N)   (sinkhorn_network.py:39-51)"""
        p = self.p
        t = F.relu(F.linear(seq[:, :, :300], p["W1_txt.weight"], p["W1_txt.bias"]))
        v = F.relu(F.linear(seq[:, :, 300:2348], p["W1_vis.weight"], p["W1_vis.bias"]))
        v = F.relu(F.linear(v, p["W2_vis.weight"], p["W2_vis.bias"]))
        x = F.relu(F.linear(torch.cat((t, v, seq[:, :, 2348:]), -1), p["W_fc_pos.weight"], p["W_fc_pos.bias"]))
        x = torch.exp(torch.tanh(F.linear(x, p["W_fc.weight"], p["W_fc.bias"])) / self.tau)
        for _ in range(self.n_iters):
            x = x / (10e-8 + x.sum(-2, keepdim=True))
            x = x / (10e-8 + x.sum(-1, keepdim=True))
        return x

    @staticmethod
    def assign(tr):
        """eval_coco.py:185-189: mx = tr^T; assignment minimising sum(max(mx) - mx[row][col]); returns col of every row (Q,N)."""
        from scipy.optimize import linear_sum_assignment
        out = []
        for m in tr.transpose(1, 2).double().numpy():
            r, c = linear_sum_assignment(m.max() - m)
            out.append(c[np.argsort(r)])
        return np.stack(out).astype(np.int64)


def assignment_gap(tr_item, n):
    """how much worse the best assignment becomes when one of the first n rows of tr^T must change its column (the columns of
    the identical all-zero padding rows count as one): the uniqueness margin of the optimum (tests skip near-ties)"""
    from scipy.optimize import linear_sum_assignment
    mx = np.asarray(tr_item, dtype=np.float64).T
    cost = mx.max() - mx
    r, c = linear_sum_assignment(cost)
    a = c[np.argsort(r)]
    tot = cost[np.arange(len(a)), a].sum()
    best2 = np.inf
    for i in range(n):
        c2 = cost.copy()
        if a[i] < n:
            c2[i, a[i]] = 1e9
        else:
            c2[i, n:] = 1e9
        r2, cc = linear_sum_assignment(c2)
        best2 = min(best2, c2[r2, cc].sum())
    return best2 - tot


def reorder_from_assignment(assign_row, locs):
    """eval_coco.py:190-200: the first len(locs) rows' assigned columns, arg-sorted, pick the slot positions."""
    sr_re = np.array([assign_row[i] for i in range(len(locs))])
    return [locs[i] for i in np.argsort(sr_re)]
