"""Index-list region format: the wire format of SURVEY 8f N2 (decode, and training when every decoder row is its own image).

The reference's callers hand the decoder a dense (rows, L, R, D) float tensor that is nothing but copies of rows of the
image's detection-feature matrix (`data/field.py:44-61`, `coco_scripts/eval_coco.py:222-247`).  Here the same information
is an int32 list `slot_idx[row, l, r]` = row of the image's feature bank (-1 = zero padding row); the HIP decoder gathers
through it (`include/vsrcap.h`: vsr_prepare_indexed), so att_va runs once per bank row instead of once per copy and no
dense tensor is ever built or shipped.

  * `IndexedRegions`         what to pass where the reference passes `det_seqs` / `det_seqs_recons` (statics[1])
  * `fill_region_indices`    `COCOControlSequenceField._fill` (field.py:44-61) producing indices instead of feature copies
  * `reorder_slots`          eval_coco.py:222-241 (slot permutation, compaction, last-slot replication, verb permutation)
                             for all captions of a batch at once, on the GPU (vsr_reorder_slots)
"""
import numpy as np
import torch


class IndexedRegions:
    """bank (n_img, Rb, D) fp32 GPU; slot_idx (B, L, R) int32 GPU, -1 = padding; row_img (B) int32 GPU or None (B == n_img)."""

    def __init__(self, bank, slot_idx, row_img=None):
        if bank.dim() != 3 or slot_idx.dim() != 3:
            raise ValueError("bank must be (n_img, Rb, D) and slot_idx (B, L, R)")
        self.bank = bank
        self.slot_idx = slot_idx.to(torch.int32)
        self.row_img = None if row_img is None else row_img.to(torch.int32)

    @property
    def device(self):
        return self.slot_idx.device

    def size(self, dim=None):
        shape = tuple(self.slot_idx.shape) + (self.bank.size(2),)
        return shape if dim is None else shape[dim]

    def to(self, device):
        return IndexedRegions(self.bank.to(device), self.slot_idx.to(device), None if self.row_img is None else self.row_img.to(device))

    def dense(self):
        """The (B, L, R, D) tensor the reference would have been given (tests; training with a row -> image map)."""
        B, L, R = self.slot_idx.shape
        img = self.row_img.long() if self.row_img is not None else torch.arange(B, device=self.slot_idx.device)
        idx = self.slot_idx.long()
        rows = self.bank[img[:, None, None].expand(B, L, R), idx.clamp(min=0)]
        return rows * (idx >= 0).unsqueeze(-1).to(rows.dtype)


def detections_inside(det_boxes, query):
    """field.py:37-43: detections whose box lies inside box `query`."""
    cond = ((det_boxes[:, 0] >= det_boxes[query, 0]) & (det_boxes[:, 1] >= det_boxes[query, 1]) &
            (det_boxes[:, 2] <= det_boxes[query, 2]) & (det_boxes[:, 3] <= det_boxes[query, 3]))
    return np.nonzero(cond)[0]


def fill_region_indices(cls_seq, n_det, det_boxes, selected_classes, most_probable_idxs, max_len, fix_length,
                        max_detections=20, all_boxes=True, sorting=False):
    """`COCOControlSequenceField._fill` (field.py:44-61) on indices: returns (fix_length, max_detections) int32, -1 = zero row.

    Gathering `det_features` through the result (zero rows for -1) reproduces `_fill`'s `det_sequences` bit for bit.
    `most_probable_idxs` is field.py:84 (`argsort(max cls prob)[::-1][:max_detections]`), `n_det = det_features.shape[0]`."""
    idx = np.full((fix_length, max_detections), -1, dtype=np.int32)
    most = np.asarray(most_probable_idxs, dtype=np.int64)
    for j, cls in enumerate(cls_seq[:max_len]):
        if cls == '_':
            n = len(idx[j, :n_det])                       # the slice the reference assigns to (:48)
            if n != len(most):
                raise ValueError("could not broadcast input array from shape (%d,) into shape (%d,)" % (len(most), n))
            idx[j, :n] = most
        else:
            seeds = [i for i, c in enumerate(selected_classes) if c == cls]
            if all_boxes:
                det_ids = np.unique(np.concatenate([detections_inside(det_boxes, d) for d in seeds]))   # raises on no seed, as :52
            else:
                det_ids = np.unique(np.asarray(seeds, dtype=np.int64))
            take = det_ids[:max_detections]
            n = len(idx[j, :len(det_ids)])
            idx[j, :n] = take
    if not sorting:
        last = len(cls_seq[:max_len])
        idx[last:] = idx[last - 1]
    return idx


def reorder_slots(engine, regions, final_ranks, verb_list=None):
    """eval_coco.py:222-241 for every caption row of `regions` at once.

    regions      IndexedRegions with slot_idx (N, L, R)
    final_ranks  list of N lists (the reference's `final_rank`, :216-221) or an (N, L) int tensor padded with -1
    verb_list    (N, L) or (N, L, 1) tensor / array of verb ids per slot (`verb_list[i]`), or None
    Returns (IndexedRegions with re-ordered slots, verbs (N, L) fp32 with -1 where the permutation has no row)."""
    N, L, R = regions.slot_idx.shape
    dev = regions.device
    if torch.is_tensor(final_ranks):
        rank = final_ranks.to(device='cpu', dtype=torch.int64).numpy()
    else:
        rank = np.full((N, L), -1, dtype=np.int64)
        for n, fr in enumerate(final_ranks):
            fr = [int(x) for x in list(fr)[:L]]           # `if j < fixed_len` (:225)
            rank[n, :len(fr)] = fr
    if rank.shape != (N, L):
        raise ValueError("final_ranks must give one rank list per caption row")
    if (rank >= L).any():                                 # perm_matrix[j, int(rk)] (:226) would raise IndexError
        raise IndexError("index %d is out of bounds for axis 1 with size %d" % (int(rank.max()), L))
    for n in range(N):                                    # a negative rk indexes from the end in numpy (:226)
        neg = (rank[n] < 0) & (np.arange(L) < _rank_len(final_ranks, n, L))
        rank[n, neg] += L
    rank_t = torch.from_numpy(rank.astype(np.int32)).to(dev)
    verbs = None
    if verb_list is not None:
        verbs = torch.as_tensor(verb_list).to(device=dev, dtype=torch.float32).reshape(N, L).contiguous()
    bank_mask = engine.row_mask(regions.bank).reshape(-1).contiguous()
    out, vout = engine.reorder_slots(regions.slot_idx.contiguous(), rank_t, verbs, bank_mask, regions.row_img, regions.bank.size(1))
    return IndexedRegions(regions.bank, out, regions.row_img), vout


def _rank_len(final_ranks, n, L):
    if torch.is_tensor(final_ranks):
        return 0                                          # tensors are already padded with -1: nothing to wrap
    return min(len(final_ranks[n]), L)
