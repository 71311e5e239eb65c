"""Eval-side batching.

coco_scripts/eval_coco.py:127-247 calls `model.beam_search_v` ONCE PER IMAGE with n_caps (~5) region orderings, i.e.
25-row problems that starve any GPU decoder.  All images of a loader batch are independent, so their caption rows can
be decoded by one call: rows of image i carry that image's pooled detections (eval_coco.py:242 expands them the same
way), their own reconstructed region sequences and verb lists."""
import torch


def beam_search_v_batched(model, items, eos_idxs, beam_size=5, out_size=1, gt=False):
    """items: list of (detections_i (R0,D), det_seqs_recons_i (n_i,L,R,D), verbs_i (n_i,L) or None).
    Returns a list with, per image, what model.beam_search_v(...) returns for it: ([words, gates], [lp_w, lp_g])."""
    dets, seqs, verbs, counts = [], [], [], []
    use_verbs = any(v is not None for _, _, v in items)
    for det_i, seq_i, verb_i in items:
        n = seq_i.size(0)
        counts.append(n)
        dets.append(det_i.unsqueeze(0).expand(n, det_i.size(0), det_i.size(1)))
        seqs.append(seq_i.float())
        if use_verbs:
            verbs.append(verb_i if verb_i is not None else torch.full((n, seq_i.size(1)), -1.0, device=seq_i.device))
    det = torch.cat(dets, 0).contiguous()
    seq = torch.cat(seqs, 0).contiguous()
    if use_verbs:
        statics = (det, seq, torch.cat([v.to(seq.device).float() for v in verbs], 0))
        outs, lps = model.beam_search_v(statics, eos_idxs=eos_idxs, beam_size=beam_size, out_size=out_size, gt=gt)
    else:
        outs, lps = model.beam_search((det, seq), eos_idxs, beam_size, out_size)
    res, lo = [], 0
    for n in counts:
        res.append(([o[lo:lo + n] for o in outs], [l[lo:lo + n] for l in lps]))
        lo += n
    return res


def beam_search_v_indexed(model, detections, bank, slot_idx, row_img, final_ranks, verb_list, eos_idxs, beam_size=5, out_size=1,
                          gt=False):
    """The whole of eval_coco.py:222-249 for a batch of images on the GPU, in the index-list region format.

    detections (n_img, R0, D) and bank (n_img, Rb, D): one entry per IMAGE (no `.expand` to n_caps rows, :242);
    slot_idx (N, L, R) int32: the slots of every caption row as rows of its image's bank (vsrcap.regions.fill_region_indices);
    row_img (N) int32: image of each caption row; final_ranks: the reference's `final_rank` per caption row (:216-221);
    verb_list (N, L): `verb_list[i][idx]` per caption row.  Returns what ONE model.beam_search_v call over all N rows returns."""
    from .regions import IndexedRegions, reorder_slots
    eng = model._engine(detections.device)
    regions = IndexedRegions(bank, slot_idx, row_img)
    regions, verbs = reorder_slots(eng, regions, final_ranks, verb_list)
    return model.beam_search_v((detections, regions, verbs), eos_idxs=eos_idxs, beam_size=beam_size, out_size=out_size, gt=gt)


# ---------------------------------------------------------------------------------------------- SURVEY 8f N4
def verb_rank_merge(la, lb):
    """Merge two slot rankings (utils/tools.py:35-71): keep la, re-order the entries lb shares with la into la's order, and
    insert lb's other entries in front of their right neighbour among the shared ones (or append)."""
    la, lb = list(la), list(lb)
    shared, pos_in_b = [], []
    for a in la:
        for j, b in enumerate(lb):
            if a == b:
                shared.append(a)
                pos_in_b.append(j)
                break
    ordered = sorted(pos_in_b)
    if ordered != pos_in_b:
        for j, pos in enumerate(ordered):
            lb[pos] = shared[j]
    right, right_of = None, {}
    for b in reversed(lb):
        if b not in shared:
            right_of[b] = right
        else:
            right = b
    merged = list(la)
    for b in lb:
        if b not in shared:
            r = right_of[b]
            if r is None:
                merged.append(b)
            else:
                merged.insert(merged.index(r), b)
    return merged


def rank_captions(ssp, sinkhorn, control_verb, det_seqs_v, det_seqs_sr, seqs_perm):
    """`final_rank` of eval_coco.py:141-221 for ALL caption rows of a loader batch, with ONE S-SSP call and ONE Sinkhorn call.

    control_verb (N, max_verb) ints (0 = none), det_seqs_v (N, L, max_verb) ints, det_seqs_sr (N, L, max_sr) ints: host arrays
    (they are integer annotations of the data loader); seqs_perm (N, L, 2352) fp32 GPU tensor: the rows the reference
    concatenates at :147 for the Sinkhorn net.  ssp / sinkhorn: models.S_SSP / models.SinkhornNet on the GPU.
    Returns a list of N rankings (lists of slot positions) for vsrcap.regions.reorder_slots / beam_search_v_indexed.
    Host part = the reference's integer bookkeeping (:148-166, :190-221); device part = every network evaluation and the assignments."""
    import numpy as np
    control_verb, det_seqs_v, det_seqs_sr = (np.asarray(x) for x in (control_verb, det_seqs_v, det_seqs_sr))
    N, L = det_seqs_v.shape[0], det_seqs_v.shape[1]
    jobs = []                                             # (caption, verb, roles (L,), sr_find, need_re_rank)
    for n in range(N):
        for verb in control_verb[n]:
            if verb == 0:
                break
            roles = np.zeros(L, dtype=np.int64)
            find_sr, sr_find, need = 0, {}, set()
            for j in range(L):
                for k in range(det_seqs_v.shape[2]):
                    if verb == det_seqs_v[n, j, k] and find_sr < 10:
                        sr = int(det_seqs_sr[n, j, k])
                        if sr not in sr_find:
                            sr_find[sr] = [j]
                            roles[find_sr] = sr
                            find_sr += 1
                        else:
                            sr_find[sr].append(j)
                            need.add(sr)
            if find_sr:
                jobs.append((n, int(verb), roles, sr_find, need))
    if not jobs:
        return [[] for _ in range(N)]
    dev = seqs_perm.device
    pred, _ = ssp.generate_batch(torch.tensor([j[1] for j in jobs], device=dev), torch.from_numpy(np.stack([j[2] for j in jobs])).to(dev))
    # Sinkhorn items: one per (job, repeated role); rows = the feature rows of the slots that carry the role, zero padded (:178-182)
    items, gather = [], []
    SN = sinkhorn.N
    for ji, (n, _, _, sr_find, need) in enumerate(jobs):
        for sr in sorted(need):
            locs = sr_find[sr][:SN]
            items.append((ji, sr, locs))
            gather.append([n * L + loc for loc in locs] + [-1] * (SN - len(locs)))
    sr_rank = {}
    if items:
        g = torch.tensor(gather, device=dev)
        rows = seqs_perm.reshape(N * L, -1).float()
        seq = rows[g.clamp(min=0)] * (g >= 0).unsqueeze(-1).to(rows.dtype)
        _, assign = sinkhorn.assign(seq.contiguous())
        assign = assign.cpu().numpy()
        for (ji, sr, locs), a in zip(items, assign):
            order = np.argsort(np.array([a[i] for i in range(len(locs))]))            # :190-200
            sr_rank[(ji, sr)] = [locs[i] for i in order]
    pred = pred.cpu().numpy()
    ranks = [[] for _ in range(N)]
    for ji, (n, _, _, sr_find, _) in enumerate(jobs):
        verb_rank = []
        for sr in pred[ji]:
            if sr == 0:
                break
            verb_rank += sr_rank[(ji, int(sr))] if len(sr_find[int(sr)]) != 1 else sr_find[int(sr)]
        ranks[n].append(verb_rank)
    out = []
    for n in range(N):
        if not ranks[n]:
            out.append([])                                 # the reference indexes verb_ranks[0] here and raises (:210)
            continue
        final = ranks[n][0]
        for other in ranks[n][1:]:
            final = verb_rank_merge(final, other)
        out.append([int(x) for x in final])
    return out
