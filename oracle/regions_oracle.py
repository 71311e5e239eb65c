"""CPU oracle for the region wire format on either side of the decoder.  TEST INFRASTRUCTURE ONLY:
imported by tests/ and tests/golden/make_golden_regions.py, never by the product path.

Dense numpy restatements of the two places where the reference builds the decoder's region tensor:

  fill_dense         COCOControlSequenceField._fill                      /root/reference/data/field.py:44-61
  reconstruct_dense  the slot re-ordering block of the beam-eval loop    /root/reference/coco_scripts/eval_coco.py:222-238

Parity pinned: tests/golden/g7_fill.npz and g8_reorder.npz hold outputs of those very statements of the reference,
executed in the build container on seeded inputs (tests/golden/make_golden_regions.py extracts the method / the block from
the reference files at run time; `data/field.py` and `eval_coco.py` cannot be imported whole because speaksee, h5py and
munkres are absent).  tests/test_regions.py checks this oracle against them bit for bit.
"""
import numpy as np


def detections_inside(det_boxes, query):                                   # field.py:37-43
    c1 = det_boxes[:, 0] >= det_boxes[query, 0]
    c2 = det_boxes[:, 1] >= det_boxes[query, 1]
    c3 = det_boxes[:, 2] <= det_boxes[query, 2]
    c4 = det_boxes[:, 3] <= det_boxes[query, 3]
    return np.nonzero(c1 & c2 & c3 & c4)[0]


def fill_dense(cls_seq, det_features, det_boxes, selected_classes, most_probable_dets, max_len, fix_length,
               max_detections, all_boxes=True, sorting=False):
    """field.py:44-61.  Returns (fix_length, max_detections, D) float32."""
    out = np.zeros((fix_length, max_detections, det_features.shape[-1]))   # :45 (float64 until the final cast)
    for j, cls in enumerate(cls_seq[:max_len]):                            # :46
        if cls == '_':
            out[j, :det_features.shape[0]] = most_probable_dets            # :48
        else:
            seeds = [i for i, c in enumerate(selected_classes) if c == cls]                               # :50
            if all_boxes:
                ids = np.unique(np.concatenate([detections_inside(det_boxes, d) for d in seeds]))         # :52
            else:
                ids = np.unique(seeds)                                     # :54
            out[j, :len(ids)] = np.take(det_features, ids, axis=0)[:max_detections]                       # :55
    if not sorting:                                                        # :57-59
        last = len(cls_seq[:max_len])
        out[last:] = out[last - 1]
    return out.astype(np.float32)                                          # :61


def reconstruct_dense(this_seqs_all, final_rank, this_verb_list, fixed_len):
    """eval_coco.py:222-238 for one caption.  this_seqs_all (fixed_len, R, D); this_verb_list (fixed_len, 1).
    Returns (recons_row (fixed_len, R, D) float64, verb_row (fixed_len, 1) float64)."""
    perm_matrix = np.zeros((fixed_len, fixed_len))                         # :223
    for j, rk in enumerate(final_rank):                                    # :224-226
        if j < fixed_len:
            perm_matrix[j, int(rk)] = 1
    flat = np.reshape(this_seqs_all, (this_seqs_all.shape[0], -1))         # :227
    recons = np.dot(perm_matrix, flat)                                     # :228
    recons = np.reshape(recons, this_seqs_all.shape)                       # :229
    recons = recons[np.sum(recons, (1, 2)) != 0]                           # :230 drop empty slots
    row = np.zeros(this_seqs_all.shape)
    last = recons.shape[0] - 1                                             # :232
    row[:recons.shape[0]] = recons                                         # :233
    row[last + 1:] = recons[last:last + 1]                                 # :234 replicate the last kept slot
    perm_mask = (np.sum(perm_matrix, -1) == 0).astype(int)                 # :237
    verb_row = -1 * perm_mask[:, np.newaxis] + np.dot(perm_matrix, this_verb_list)                        # :238
    return row, verb_row


def gather_dense(bank, slot_idx):
    """Dense tensor an index list stands for: bank (Rb, D), slot_idx (L, R) with -1 = zero row."""
    out = np.zeros(slot_idx.shape + (bank.shape[-1],), dtype=bank.dtype)
    live = slot_idx >= 0
    out[live] = bank[slot_idx[live]]
    return out
