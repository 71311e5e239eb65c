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
