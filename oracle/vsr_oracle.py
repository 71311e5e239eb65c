"""CPU oracle for the VSR-guided captioning decoder hot path.  TEST INFRASTRUCTURE ONLY.

This file is the parity checker, not the product: only tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py may import it.  The product path (vsr-guided-cic_amd/) never does and
fails loudly when its HIP library is missing.

It is an independent restatement (PyTorch CPU, fp32 or fp64) of the algorithm in
  /root/reference/models/controllable_captioning.py   (step :117-190, step_v :192-297, init_state :109-115)
  /root/reference/models/CaptioningModel.py           (forward :22-36, test :38-52, sample_rl :54-76,
                                                       beam_search :116-195, beam_search_v :197-294)
written from the equations of SURVEY.md section 8a, not from the reference's text.  Parity is PINNED:
tests/golden/make_golden.py imports the real reference in the build container and tests/test_oracle.py
checks this restatement against those committed vectors (and, when /root/reference is mounted, against the
reference live) - token ids exactly, log-probs to 1e-5 (fp32) / 1e-10 (fp64).

Two flavours of the same arithmetic:
  as_written=True   the reference's cost profile: pooled descriptor, region projection and statics
                    re-gather are redone every timestep, beam search does a full sort.  This is what
                    bench.py times as the "reference CPU path" (cpu_baseline.kind == "port").
  as_written=False  hoisted: per-image work done once, beams index their image, top-k instead of sort.
Both give identical tokens; log-probs agree to rounding.
"""
import torch
import torch.nn.functional as F


def _lstm(x_ih, x_hh, c_prev, H):
    """PyTorch LSTMCell gate order i, f, g, o (controllable_captioning.py:152,177 use nn.LSTMCell)."""
    pre = x_ih + x_hh
    i, f, g, o = pre[:, :H], pre[:, H:2 * H], pre[:, 2 * H:3 * H], pre[:, 3 * H:]
    c = torch.sigmoid(f) * c_prev + torch.sigmoid(i) * torch.tanh(g)
    h = torch.sigmoid(o) * torch.tanh(c)
    return h, c


class Oracle:
    def __init__(self, params, seq_len, bos_idx, h2_first_lstm=True, img_second_lstm=False,
                 verb_table=None, as_written=True, dtype=torch.float32):
        """params: dict state_dict-name -> array/tensor in the reference layout (A0 in SURVEY.md 8a)."""
        self.p = {k: torch.as_tensor(v).to(dtype) for k, v in params.items()}
        self.seq_len = seq_len
        self.bos_idx = bos_idx
        self.h2_first = h2_first_lstm
        self.img_second = img_second_lstm
        self.verb_table = verb_table or {}
        self.as_written = as_written
        self.dtype = dtype
        self.H = self.p["W1_hs.weight"].shape[0]
        self.V = self.p["out_fc.weight"].shape[0]
        self._hoist = None

    # ------------------------------------------------------------------ state
    def init_state(self, n):
        """controllable_captioning.py:109-115 (zeros; slot pointer int64)."""
        z = lambda: torch.zeros(n, self.H, dtype=self.dtype)
        return [z(), z(), z(), z(), torch.zeros(n, dtype=torch.long)]

    # ------------------------------------------------------------------ hoisted per-image work
    def _pooled(self, det):
        """mean over the non-zero rows of det (controllable_captioning.py:126-128)."""
        valid = (det.sum(-1, keepdim=True) != 0).to(det.dtype)
        return det.sum(1) / valid.sum(1)

    def prepare(self, det, ctrl):
        """hoisted flavour: per-image descriptor and per-(image,slot) region projections/masks."""
        p = self.p
        self._hoist = {
            "vbar": self._pooled(det),
            "proj": ctrl @ p["att_va.weight"].t(),                      # (B, L, R, A)
            "mask": (ctrl.sum(-1) != 0).to(ctrl.dtype),                  # (B, L, R)
        }

    # ------------------------------------------------------------------ one timestep (A1 / A2)
    def step(self, t, state, prev, det, ctrl, seqs=None, mode="feedback", img_of_row=None,
             verbs=None, gt=False):
        """One decoder timestep.  det (N,R0,D) / ctrl (N,L,R,D) are per ROW in the as-written flavour
        and per IMAGE (with img_of_row) in the hoisted one.  Returns ((logp_word, logp_gate), state)."""
        assert mode in ("teacher_forcing", "feedback")
        p, H = self.p, self.H
        h1, c1, h2, c2, k = state
        n = h1.shape[0]
        rows = torch.arange(n)
        img = rows if img_of_row is None else img_of_row

        if mode == "teacher_forcing":
            w_prev = seqs[0][:, t]
            regions = seqs[1][:, t]                                     # (N, R, D)
            proj = mask = None
        else:
            if t == 0:
                w_prev = torch.full((n,), self.bos_idx, dtype=torch.long)
            else:
                w_prev = prev[0]
                k = torch.clamp(k + prev[1], 0, ctrl.shape[1] - 1)
            regions = ctrl[img, k]                                      # gather of the current slot
            if self._hoist is not None and not self.as_written:
                proj = self._hoist["proj"][img, k]
                mask = self._hoist["mask"][img, k]
            else:
                proj = mask = None

        if self._hoist is not None and not self.as_written:
            vbar = self._hoist["vbar"][img]
        else:
            vbar = self._pooled(det[img] if img_of_row is not None else det)

        x = p["embed.weight"][w_prev]
        in1 = torch.cat([h2, vbar, x], 1) if self.h2_first else torch.cat([vbar, x], 1)

        # sentinel gate uses the OLD h1, the shift gate (below) the NEW h1
        s_gate = torch.sigmoid(in1 @ p["W1_is.weight"].t() + p["W1_is.bias"]
                               + h1 @ p["W1_hs.weight"].t() + p["W1_hs.bias"])
        h1, c1 = _lstm(in1 @ p["lstm_cell_1.weight_ih"].t() + p["lstm_cell_1.bias_ih"],
                       h1 @ p["lstm_cell_1.weight_hh"].t() + p["lstm_cell_1.bias_hh"], c1, H)
        tc1 = torch.tanh(c1)
        s_t = s_gate * tc1
        sentinel = s_t @ p["s_fc.weight"].t() + p["s_fc.bias"]          # (N, D)

        hA = h1 @ p["att_ha.weight"].t()                                # (N, A)
        if proj is None:
            proj = regions @ p["att_va.weight"].t()                     # (N, R, A)
        if mask is None:
            mask = (regions.sum(-1) != 0).to(regions.dtype)             # (N, R)
        z_det = torch.tanh(proj + hA[:, None, :]) @ p["att_a.weight"][0]        # (N, R)
        z_sent = torch.tanh(s_t @ p["att_sa.weight"].t() + hA) @ p["att_s.weight"][0]   # (N,)
        m0 = (sentinel.sum(-1) != 0).to(regions.dtype)
        m_all = torch.cat([m0[:, None], mask], 1)                       # (N, R+1)
        alpha = F.softmax(torch.cat([z_sent[:, None], z_det], 1), 1) * m_all
        alpha = alpha / alpha.sum(1, keepdim=True)
        att = alpha[:, 0:1] * sentinel + (alpha[:, 1:, None] * regions).sum(1)  # (N, D)

        in2 = [h1, att] + ([vbar] if self.img_second else [])
        h2, c2 = _lstm(torch.cat(in2, 1) @ p["lstm_cell_2.weight_ih"].t() + p["lstm_cell_2.bias_ih"],
                       h2 @ p["lstm_cell_2.weight_hh"].t() + p["lstm_cell_2.bias_hh"], c2, H)
        logp_w = F.log_softmax(h2 @ p["out_fc.weight"].t() + p["out_fc.bias"], -1)

        g_gate = torch.sigmoid(in1 @ p["W1_ig.weight"].t() + p["W1_ig.bias"]
                               + h1 @ p["W1_hg.weight"].t() + p["W1_hg.bias"])
        g_t = g_gate * tc1
        z_g = torch.tanh(g_t @ p["att_ga.weight"].t() + hA) @ p["att_g.weight"][0]
        shift = (mask * z_det).sum(1)                                   # RAW logits of the valid regions
        logp_g = F.log_softmax(torch.stack([z_g, shift], 1), 1)

        if verbs is not None:                                           # step_v (:192-297), feedback only
            assert mode == "feedback"
            logp_w, logp_g = self._force_verbs(logp_w, logp_g, verbs[img, k], gt)
        return (logp_w, logp_g), [h1, c1, h2, c2, k]

    def _force_verbs(self, logp_w, logp_g, verb_curr, gt):
        """Rows whose current slot carries a verb emit exactly one word (log-prob 0, the rest -1e6) and are
        forced to shift (gate = [-1e3, 0]); controllable_captioning.py:268-295."""
        verb_curr = verb_curr.long()
        logp_w = logp_w.clone()
        logp_g = logp_g.clone()
        for i in torch.nonzero(verb_curr != -1).flatten().tolist():
            v = int(verb_curr[i])
            if gt:
                pick = v
            else:
                cands = self.verb_table.get(str(v), [])
                if len(cands) == 0:
                    pick = 0
                else:
                    best, pick = -1e6, -1
                    for c in cands:                                     # strict '>' : first maximum wins
                        if logp_w[i, c] > best:
                            best, pick = float(logp_w[i, c]), c
            logp_w[i] = -1e6
            logp_w[i, pick] = 0.0
            logp_g[i, 0] = -1e3
            logp_g[i, 1] = 0.0
        return logp_w, logp_g

    # ------------------------------------------------------------------ A4: teacher-forced unroll
    def forward(self, det, captions, ctrl_seq):
        """CaptioningModel.py:22-36 -> (B,T,V) word log-probs, (B,T,2) gate log-probs."""
        state = self.init_state(det.shape[0])
        outs_w, outs_g = [], []
        saved, self._hoist = self._hoist, None
        for t in range(captions.shape[1]):
            (lw, lg), state = self.step(t, state, None, det, None, (captions, ctrl_seq), "teacher_forcing")
            outs_w.append(lw)
            outs_g.append(lg)
        self._hoist = saved
        return torch.stack(outs_w, 1), torch.stack(outs_g, 1)

    # ------------------------------------------------------------------ A5: greedy
    def test(self, det, ctrl, verbs=None, gt=False, return_trace=False):
        """CaptioningModel.py:38-52: independent arg-max of word and gate, fed back; always T steps."""
        if not self.as_written:
            self.prepare(det, ctrl)
        state = self.init_state(det.shape[0])
        prev, W, G, margins, ks = None, [], [], [], []
        for t in range(self.seq_len):
            (lw, lg), state = self.step(t, state, prev, det, ctrl, verbs=verbs, gt=gt)
            prev = (lw.argmax(-1), lg.argmax(-1))
            W.append(prev[0])
            G.append(prev[1])
            if return_trace:
                top2 = torch.topk(lw, 2, -1)[0]
                margins.append(torch.stack([top2[:, 0] - top2[:, 1], (lg[:, 0] - lg[:, 1]).abs()], 1))
                ks.append(state[4])
        out = (torch.stack(W, 1), torch.stack(G, 1))
        if return_trace:
            return out + (torch.stack(margins, 1), torch.stack(ks, 1), state)
        return out

    # ------------------------------------------------------------------ A6: sampling
    def sample_rl(self, det, ctrl, forced=None, generator=None):
        """CaptioningModel.py:54-76.  forced=(words, gates) replays given samples (parity by replay,
        SURVEY.md 8c G5); otherwise draws with torch.multinomial like Categorical.sample()."""
        if not self.as_written:
            self.prepare(det, ctrl)
        state = self.init_state(det.shape[0])
        prev, W, G, LW, LG = None, [], [], [], []
        for t in range(self.seq_len):
            (lw, lg), state = self.step(t, state, prev, det, ctrl)
            if forced is not None:
                w, g = forced[0][:, t], forced[1][:, t]
            else:
                w = torch.multinomial(lw.exp(), 1, generator=generator)[:, 0]
                g = torch.multinomial(lg.exp(), 1, generator=generator)[:, 0]
            prev = (w, g)
            W.append(w); G.append(g)
            LW.append(lw.gather(1, w[:, None])[:, 0]); LG.append(lg.gather(1, g[:, None])[:, 0])
        return (torch.stack(W, 1), torch.stack(G, 1)), (torch.stack(LW, 1), torch.stack(LG, 1))

    # ------------------------------------------------------------------ A7/A8: joint (word x gate) beam search
    def beam_search(self, det, ctrl, eos_idxs, beam_size, out_size=1, verbs=None, gt=False,
                    return_scores=False):
        """CaptioningModel.py:116-195 (and :197-294 when verbs is given).

        Candidate score = seq_lp + (logp_word + logp_gate), in that association order, over
        (current beams) x V x 2; the best `beam_size` per image survive.  Finished hypotheses are frozen
        only when EVERY output stream has hit its EOS (with eos_idxs=[eos,-1] that never happens, so this
        is fixed-length beam search); the per-stream masks only zero the RETURNED log-probs, which follow
        beam slots, not ancestry."""
        B, V, T = det.shape[0], self.V, self.seq_len
        if not self.as_written:
            self.prepare(det, ctrl)
        state = self.init_state(B)
        cb = 1
        seq_lp = torch.zeros(B, 1, dtype=self.dtype)
        masks = [torch.ones(B, beam_size, dtype=self.dtype) for _ in range(2)]
        hist = [torch.zeros(B, beam_size, 0, dtype=torch.long) for _ in range(2)]
        lps = [[], []]
        prev = None
        img_of_row = torch.arange(B)
        det_r, ctrl_r, verbs_r = det, ctrl, verbs
        for t in range(T):
            if self.as_written:        # per-row statics, re-gathered after every selection like the reference
                outs, state = self.step(t, state, prev, det_r, ctrl_r, verbs=verbs_r, gt=gt)
            else:
                outs, state = self.step(t, state, prev, det, ctrl, img_of_row=img_of_row, verbs=verbs, gt=gt)
            lw = outs[0].view(B, cb, V)
            lg = outs[1].view(B, cb, 2)
            cand = seq_lp[:, :, None, None] + (lw[:, :, :, None] + lg[:, :, None, :])
            if t > 0:
                for s, eos in enumerate(eos_idxs):
                    masks[s] = masks[s] * (prev[s].view(B, cb) != eos).to(self.dtype)
                lw_ret = lw * masks[0][:, :, None]
                lg_ret = lg * masks[1][:, :, None]
                frozen = seq_lp[:, :, None, None].expand_as(cand).clone()
                frozen[:, :, 1:] = -999
                alive = torch.clamp(masks[0] + masks[1], 0, 1)[:, :, None, None]
                cand = alive * cand + frozen * (1 - alive)
            else:
                lw_ret, lg_ret = lw, lg
            flat = cand.reshape(B, -1)
            if self.as_written:
                top_v, top_i = torch.sort(flat, -1, descending=True)
                top_v, top_i = top_v[:, :beam_size], top_i[:, :beam_size]
            else:
                top_v, top_i = torch.topk(flat, beam_size, -1)
            parent = top_i // (2 * V)
            word = (top_i % (2 * V)) // 2
            gate = top_i % 2

            rows = (torch.arange(B)[:, None] * cb + parent).reshape(-1)
            state = [s[rows] for s in state]
            if self.as_written:
                det_r, ctrl_r = det_r[rows], ctrl_r[rows]
                if verbs_r is not None:
                    verbs_r = verbs_r[rows]
            img_of_row = torch.arange(B).repeat_interleave(beam_size)
            if t > 0:
                masks = [m.gather(1, parent) for m in masks]
                hist = [h.gather(1, parent[:, :, None].expand(-1, -1, h.shape[2])) for h in hist]
            hist = [torch.cat([hist[0], word[:, :, None]], 2), torch.cat([hist[1], gate[:, :, None]], 2)]
            pw = lw_ret.gather(1, parent[:, :, None].expand(-1, -1, V)).gather(2, word[:, :, None])[:, :, 0]
            pg = lg_ret.gather(1, parent[:, :, None].expand(-1, -1, 2)).gather(2, gate[:, :, None])[:, :, 0]
            lps[0].append(pw); lps[1].append(pg)
            seq_lp = top_v
            prev = (word.reshape(-1), gate.reshape(-1))
            cb = beam_size

        final, order = torch.sort(seq_lp, 1, descending=True)
        idx = order[:, :, None].expand(-1, -1, T)
        outs = [h.gather(1, idx)[:, :out_size] for h in hist]
        logs = [torch.stack(l, 2).gather(1, idx)[:, :out_size] for l in lps]
        if out_size == 1:
            outs = [o[:, 0] for o in outs]
            logs = [l[:, 0] for l in logs]
        if return_scores:
            return outs, logs, final[:, :out_size]
        return outs, logs


# ---------------------------------------------------------------------- caller arithmetic (C1, C2)
def xe_loss(logp_w, logp_g, captions, gate_gts):
    """coco_scripts/train.py:106-110: NLL(out[:, :-1], captions[:, 1:]) + 4 * NLL_ignore(-1)(gate, gts)."""
    V = logp_w.shape[-1]
    loss_cap = F.nll_loss(logp_w[:, :-1].reshape(-1, V), captions[:, 1:].reshape(-1))
    loss_gate = F.nll_loss(logp_g.reshape(-1, 2), gate_gts.reshape(-1).long(), ignore_index=-1)
    return loss_cap + 4 * loss_gate, loss_cap, loss_gate


def scst_loss(lp_w, lp_g, reward, baseline):
    """coco_scripts/train.py:174-175: -(mean_t lp_word + mean_t lp_gate) * (r - r_b), mean over the batch."""
    return (-(lp_w.mean(-1) + lp_g.mean(-1)) * (reward - baseline)).mean()
