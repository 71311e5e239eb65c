"""Golden vectors for the region wire format: outputs of the REFERENCE's own statements on seeded inputs.

    python tests/golden/make_golden_regions.py          (build container only: reads /root/reference)

`data/field.py` and `coco_scripts/eval_coco.py` cannot be imported (speaksee / h5py / munkres are absent), so the two
pieces that build the decoder's region tensor are taken out of the files at run time and executed as they stand:
  g7_fill.npz     COCOControlSequenceField._fill + get_detections_inside   (field.py:37-61, located with ast)
  g8_reorder.npz  the slot re-ordering statements of the beam-eval loop    (eval_coco.py:222-238, located by their first/last line)
Only inputs and outputs are stored; nothing of the reference's text is.
"""
import ast
import os
import sys
import textwrap

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def reference_fill():
    src = open(os.path.join(REF, "data", "field.py")).read()
    tree = ast.parse(src)
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "COCOControlSequenceField"][0]
    ns = {"np": np}
    for n in cls.body:
        if isinstance(n, ast.FunctionDef) and n.name in ("_fill", "get_detections_inside"):
            exec(textwrap.dedent(ast.get_source_segment(src, n)), ns)

    class Field:
        get_detections_inside = ns["get_detections_inside"]
        _fill = ns["_fill"]

        def __init__(self, fix_length, max_detections, all_boxes, sorting):
            self.fix_length, self.max_detections, self.all_boxes, self.sorting = fix_length, max_detections, all_boxes, sorting
    return Field


def reference_reorder_block():
    lines = open(os.path.join(REF, "coco_scripts", "eval_coco.py")).read().split("\n")
    a = [i for i, l in enumerate(lines) if "perm_matrix = np.zeros((fixed_len, fixed_len))" in l][0]
    b = [i for i, l in enumerate(lines) if "img_verb_list[idx] = " in l][0]
    return compile(textwrap.dedent("\n".join(lines[a:b + 1])), "eval_coco.py:%d-%d" % (a + 1, b + 1), "exec")


def fill_cases():
    rng = np.random.RandomState(7)
    cases = []
    for (n_det, fix_length, max_det, all_boxes, sorting, n_cls, seq_len, max_len) in [
            (12, 8, 6, True, False, 4, 5, 6), (12, 8, 6, False, False, 4, 5, 6), (30, 10, 20, True, False, 6, 7, 8),
            (5, 6, 20, True, False, 3, 4, 4), (30, 10, 4, True, False, 3, 9, 8), (20, 8, 20, True, True, 5, 4, 6),
            (9, 8, 5, False, False, 2, 8, 6), (16, 6, 8, True, False, 4, 1, 4)]:
        D = 8
        feats = np.maximum(rng.rand(n_det, D) - 0.3, 0.0)
        xy = rng.rand(n_det, 2) * 0.6
        wh = 0.05 + rng.rand(n_det, 2) * 0.4
        boxes = np.concatenate([xy, xy + wh], 1)
        boxes[0] = [0.0, 0.0, 1.0, 1.0]                       # one box that contains all the others
        sel = rng.randint(0, n_cls, size=n_det)
        seq = [int(sel[rng.randint(n_det)]) if rng.rand() > 0.25 else -1 for _ in range(seq_len)]   # -1 = the '_' class
        probs = rng.rand(n_det)
        most_idx = np.argsort(probs)[::-1][:max_det]          # field.py:84
        cases.append(dict(feats=feats, boxes=boxes, sel=sel, seq=np.array(seq), most_idx=most_idx,
                          params=np.array([fix_length, max_det, int(all_boxes), int(sorting), max_len])))
    return cases


def reorder_cases():
    rng = np.random.RandomState(11)
    cases = []
    for (L, R, Rb, kind) in [(10, 4, 12, "perm"), (10, 4, 12, "short"), (8, 3, 9, "dup"), (10, 5, 20, "empty"),
                             (6, 4, 10, "long"), (10, 4, 12, "single"), (10, 20, 36, "perm"), (10, 4, 12, "tail")]:
        D = 8
        bank = np.maximum(rng.rand(Rb, D) - 0.3, 0.0).astype(np.float32)
        bank[1] = 0.0                                          # an all-zero bank row (masked like padding)
        n_slots = rng.randint(2, L + 1)
        idx = np.full((L, R), -1, dtype=np.int32)
        for l in range(n_slots):
            n = rng.randint(1, R + 1)
            idx[l, :n] = np.sort(rng.choice(Rb, n, replace=False))
        idx[n_slots:] = idx[n_slots - 1]                       # _fill's tail replication
        if kind == "empty":
            idx[1] = -1                                        # a slot with no rows
            idx[2, :] = -1
            idx[2, 0] = 1                                      # a slot whose only row is the all-zero bank row
        if kind == "perm":
            rank = list(rng.permutation(n_slots))
        elif kind == "short":
            rank = list(rng.permutation(n_slots))[:max(1, n_slots // 2)]
        elif kind == "dup":
            rank = [0, 1, 1, 0][:n_slots] + [0]
        elif kind == "empty":
            rank = list(rng.permutation(min(n_slots, L)))
            if 1 not in rank:
                rank[0] = 1
        elif kind == "long":
            rank = list(rng.permutation(L)) + [0, 1, 2]        # longer than fixed_len: the j < fixed_len guard
        elif kind == "single":
            rank = [n_slots - 1]
        else:
            rank = list(range(L))                              # identity over the replicated tail
        verbs = np.where(rng.rand(L, 1) > 0.6, rng.randint(0, 50, size=(L, 1)), -1).astype(np.float64)
        cases.append(dict(bank=bank, idx=idx, rank=np.array(rank, dtype=np.int64), verbs=verbs))
    return cases


def main():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
    import regions_oracle as ro
    Field = reference_fill()
    out = {}
    for i, c in enumerate(fill_cases()):
        fix_length, max_det, all_boxes, sorting, max_len = [int(x) for x in c["params"]]
        names = ["_" if s < 0 else "c%d" % s for s in c["seq"]]
        sel_names = ["c%d" % s for s in c["sel"]]
        f = Field(fix_length, max_det, bool(all_boxes), bool(sorting))
        got = f._fill(names, c["feats"], c["boxes"], sel_names, c["feats"][c["most_idx"]], max_len)
        for k, v in c.items():
            out["c%d_%s" % (i, k)] = v
        out["c%d_out" % i] = got
    out["n"] = np.array(len(fill_cases()))
    np.savez_compressed(os.path.join(HERE, "g7_fill.npz"), **out)
    print("g7_fill.npz: %d cases" % int(out["n"]))

    block = reference_reorder_block()
    out = {}
    for i, c in enumerate(reorder_cases()):
        L = c["idx"].shape[0]
        dense = ro.gather_dense(c["bank"], c["idx"])
        env = dict(np=np, fixed_len=L, final_rank=list(c["rank"]), this_seqs_all=dense, this_verb_list=c["verbs"],
                   det_seqs_recons=np.zeros((1,) + dense.shape), img_verb_list=np.zeros((1, L, 1)), idx=0)
        exec(block, env)
        for k, v in c.items():
            out["c%d_%s" % (i, k)] = v
        out["c%d_recons" % i] = env["det_seqs_recons"][0]
        out["c%d_verbs_out" % i] = env["img_verb_list"][0]
    out["n"] = np.array(len(reorder_cases()))
    np.savez_compressed(os.path.join(HERE, "g8_reorder.npz"), **out)
    print("g8_reorder.npz: %d cases" % int(out["n"]))


if __name__ == "__main__":
    main()
