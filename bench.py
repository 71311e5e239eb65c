#!/usr/bin/env python3
"""bench.py - headline benchmark of the VSR captioning decoder hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload beam5|greedy] [--no-cpu]

Workload (BASELINE.json metric, configs[2]): beam-5 decode through ControllableCaptioningModel.beam_search,
batch 100 images per GPU, 36 regions x 2048-d, 10 slots, seq_len 20, vocab 10 000, fp32 (the reference's
precision; token parity holds in this mode).  A "step" = ONE full decode call on one batch of synthetic
inputs already resident in HBM: hoisted statics (vsr_prepare) + 20 timesteps + back-tracking.
tokens/s = n_gpus * B * T * steps / wall time (top-1 hypothesis tokens).
Multi-GPU: images are independent, each rank decodes its own batch with its own weight replica, no
data-path collective (weak scaling); only the timing is reduced (MAX over ranks).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "vsr-guided-cic_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from vsrcap import synth  # noqa: E402

CFG = dict(V=10000, B=100, R0=36, R=36, D=2048, L=10, T=20, E=1000, H=1000, A=512)
BEAM = 5
EOS = 3
PROFILE_EVERY = 5
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, spec


def measured_traffic():
    """HBM bytes per GEMM launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on this same
    command, FETCH_SIZE doubled per the guide's gfx950 correction); None when no profile has been committed."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    if os.path.isdir(pdir):
        for f in sorted(os.listdir(pdir)):
            if f.endswith("gemm_hbm_traffic.json"):
                try:
                    best = json.load(open(os.path.join(pdir, f)))["hbm_bytes_per_launch"]
                except Exception:
                    pass
    return best


def cpu_baseline(weights, sample_B, beam):
    """The CPU oracle in its as-written flavour (the reference's cost profile: per-step recompute of the pooled
    descriptor / region projection, statics re-gather per beam step, full sort) on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import vsr_oracle as vo
    c = CFG
    o = vo.Oracle(weights, c["T"], 2, as_written=True)
    det = torch.from_numpy(synth.make_detections(sample_B, c["R0"], c["D"], seed=77))
    ctrl = torch.from_numpy(synth.make_ctrl(sample_B, c["L"], c["R"], c["D"], seed=77))
    with torch.no_grad():
        if beam > 1:
            o.beam_search(det[:2], ctrl[:2], [EOS, -1], beam, 1)          # warm-up
            t0 = time.time()
            o.beam_search(det, ctrl, [EOS, -1], beam, 1)
        else:
            o.test(det[:2], ctrl[:2])
            t0 = time.time()
            o.test(det, ctrl)
        dt = time.time() - t0
    return dict(value=sample_B * c["T"] / dt, unit="tokens/s", cores=torch.get_num_threads(), kind="port",
                sample="oracle/vsr_oracle.py as_written, %s, %d images x %d steps, fp32, %.1f s" %
                       ("beam-%d" % beam if beam > 1 else "greedy", sample_B, c["T"], dt))


def cpu_baseline_xe(weights, sample_B):
    """XE step of coco_scripts/train.py:103-113 on the CPU oracle (autograd backward + Adam), bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import vsr_oracle as vo
    c = CFG
    o = vo.Oracle(weights, c["T"], 2, as_written=True)
    params = [o.p[k].requires_grad_(True) for k in o.p]
    opt = torch.optim.Adam(params, lr=5e-4)
    det = torch.from_numpy(synth.make_detections(sample_B, c["R0"], c["D"], seed=77))
    seq = torch.from_numpy(synth.make_ctrl(sample_B, c["T"], c["R"], c["D"], seed=78))
    caps = torch.from_numpy(synth.make_captions(sample_B, c["T"], c["V"], seed=77))
    gts = torch.from_numpy(synth.make_gate_gts(sample_B, c["T"], seed=77))
    ts = []
    for _ in range(2):
        t0 = time.time()
        opt.zero_grad()
        out, gate = o.forward(det, caps, seq)
        vo.xe_loss(out, gate, caps, gts)[0].backward()
        opt.step()
        ts.append(time.time() - t0)
    return dict(value=sample_B / ts[-1], unit="samples/s", cores=torch.get_num_threads(), kind="port",
                sample="oracle/vsr_oracle.py as_written XE step (forward + NLL losses + autograd backward + Adam), %d images x %d steps, "
                       "fp32, %.1f s (second of two steps)" % (sample_B, c["T"], ts[-1]))


def train_bench(args):
    """XE step (BASELINE configs[3]) / SCST step (configs[4]) in fp32: forward + losses + hand-written BPTT backward +
    Adam, data-parallel over ranks with RCCL gradient all-reduce and global loss normalisation (vsrcap/parallel.py)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from models import ControllableCaptioningModel
    from vsrcap import parallel
    c = CFG
    weights = synth.make_weights(c["V"], c["D"], c["E"], c["H"], c["A"], seed=0, gains={k: 1.0 for k in synth.DEFAULT_GAINS})
    m = ControllableCaptioningModel(c["T"], c["V"], 2, det_feat_size=c["D"], input_encoding_size=c["E"], rnn_size=c["H"],
                                    att_size=c["A"], verb_2_vob_all={})
    m.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    m = m.to(dev).train()
    opt = torch.optim.Adam(m.parameters(), lr=5e-4, fused=True)      # train.py:77 Adam(lr=5e-4); one fused launch per step instead of ~40 foreach kernels
    xe = args.workload == "xe"
    L = c["T"] if xe else c["L"]
    batches = []
    for i in range(2):
        seed = 2000 + 10 * rank + i
        batches.append((torch.from_numpy(synth.make_detections(c["B"], c["R0"], c["D"], seed=seed)).to(dev),
                        torch.from_numpy(synth.make_ctrl(c["B"], L, c["R"], c["D"], seed=seed)).to(dev),
                        torch.from_numpy(synth.make_captions(c["B"], c["T"], c["V"], seed=seed)).to(dev),
                        torch.from_numpy(synth.make_gate_gts(c["B"], c["T"], seed=seed)).to(dev)))
    step = parallel.DataParallelStep(list(m.parameters()), opt, forward_fn=lambda d, cp, sq: m((d,), (cp, sq)),
                                     sample_fn=lambda d, ct: m.sample_rl(d, ct))
    NS = 5                          # samples per image (BASELINE configs[4]); the reference has no such loop, the caller
    rl_batches = []                 # repeats every image NS times
    if not xe:
        for det, reg, _, _ in batches:
            rl_batches.append((det.repeat_interleave(NS, 0).contiguous(), reg.repeat_interleave(NS, 0).contiguous()))

    # rewards: per-sample CIDEr-D on the device (vsrcap/reward.py; train.py:169-170) of the sampled / greedy
    # captions against the batch's synthetic reference caption, document frequencies from a synthetic corpus of 2 000 captions
    if not xe:
        from vsrcap.reward import CiderD, clean_ids
        corpus = [[clean_ids(cap, eos=EOS)] for cap in synth.make_captions(2000, c["T"], c["V"], seed=77)]
        cider = CiderD(corpus, c["V"])
        refs = [caps.unsqueeze(1).contiguous() for _, _, caps, _ in batches]               # (B, 1, T): one reference per sample
        refs5 = [r.repeat_interleave(NS, 0).contiguous() for r in refs]

    def one_step(i):
        det, reg, caps, gts = batches[i & 1]
        if xe:
            return step.xe_step(det, caps, reg, gts)
        with torch.no_grad():       # greedy baseline of train.py:127-138 (model.test), then NS samples per image
            m.eval()
            base_words, _ = m.test(det, reg)
            m.train()
        r_base = cider.rewards(base_words, refs[i & 1], EOS).repeat_interleave(NS, 0)
        det5, reg5 = rl_batches[i & 1]
        return step.scst_step(det5, reg5, lambda words: (cider.rewards(words, refs5[i & 1], EOS), r_base))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        one_step(i)
    eng = m._engine(dev)
    barrier()
    eng.profile_begin(every=PROFILE_EVERY)
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(i)
    barrier()
    dt = time.perf_counter() - t0
    gemm_seen = eng.profile_seen()
    gemm_ms, gemm_n, gemm_flops = eng.profile_end(dev)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        line = {
            "metric": "XE-step samples/sec" if xe else "SCST-step images/sec (5 samples/image + greedy baseline, CIDEr-D rewards on device)",
            "value": world * c["B"] * args.steps / dt, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("XE training step (forward + NLL losses + BPTT backward + torch.optim.Adam(fused=True)), batch 100/GPU, 20 slots x 36 regions x 2048-d, "
                                    "seq_len 20, vocab 10000 (BASELINE configs[3], fp32)") if xe else
                                   ("SCST step: greedy baseline (100 images) + sample_rl on 500 rows (5 samples/image) + replayed forward + BPTT "
                                    "backward + Adam(fused=True), rewards = device CIDEr-D vs synthetic references, 10 slots x 36 x 2048 (BASELINE configs[4], fp32)"),
                       "batch_per_gpu": c["B"], "seq_len": c["T"], "parallelism": "dp%d, RCCL gradient all-reduce" % world},
            "roofline": {"bound": "mfma", "kernel": "gemm_nt_f32_kernel (v_mfma_f32_32x32x2_f32)", "achieved": achieved,
                         "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                         "launches": gemm_seen, "launches_timed": gemm_n, "avg_launch_us": gemm_ms * 1e3 / max(gemm_n, 1),
                         "gemm_share_of_wall": gemm_ms * 1e-3 / max(gemm_n, 1) * gemm_seen / dt},
        }
        if world == 1 and not args.no_cpu and xe:
            line["cpu_baseline"] = cpu_baseline_xe(weights, min(args.cpu_sample, 16))
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="beam5", choices=["beam5", "beam5idx", "greedy", "xe", "scst"])
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=24)
    args = ap.parse_args()

    if args.workload in ("xe", "scst"):
        return train_bench(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from models import ControllableCaptioningModel
    c = CFG
    weights = synth.make_weights(c["V"], c["D"], c["E"], c["H"], c["A"], seed=0)
    m = ControllableCaptioningModel(c["T"], c["V"], 2, det_feat_size=c["D"], input_encoding_size=c["E"], rnn_size=c["H"],
                                    att_size=c["A"], verb_2_vob_all={})
    m.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    m = m.to(dev).eval()
    # two distinct resident batches per rank, alternated, so no step can reuse the previous step's prepare()
    batches = []
    for i in range(2):
        seed = 1000 + 10 * rank + i
        batches.append((torch.from_numpy(synth.make_detections(c["B"], c["R0"], c["D"], seed=seed)).to(dev),
                        torch.from_numpy(synth.make_ctrl(c["B"], c["L"], c["R"], c["D"], seed=seed)).to(dev)))
    beam = BEAM if args.workload in ("beam5", "beam5idx") else 1
    indexed = args.workload == "beam5idx"
    if indexed:
        # index-list region format: the slots name rows of the image's own detection matrix instead of
        # carrying copies of them; same shapes as the headline workload, decoded through vsr_prepare_indexed
        from vsrcap.regions import IndexedRegions
        batches = []
        for i in range(2):
            seed = 1000 + 10 * rank + i
            det = torch.from_numpy(synth.make_detections(c["B"], c["R0"], c["D"], seed=seed, min_valid=c["R0"])).to(dev)
            idx = torch.from_numpy(synth.make_slot_indices(c["B"], c["L"], c["R"], c["R0"], seed=seed)).to(dev)
            batches.append((det, IndexedRegions(det, idx)))

    def one_step(i):
        det, ctrl = batches[i & 1]
        with torch.no_grad():
            if beam > 1:
                return m.beam_search((det, ctrl), [EOS, -1], beam, 1)
            return m.test(det, ctrl)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        one_step(i)
    eng = m._engine(dev)
    barrier()
    # HIP events around every 5th GEMM launch of the timed region (3 launch kinds per timestep: every kind is sampled
    # equally often); an event pair on EVERY launch costs the timed region 3 % (192.3 k vs 198.7 k tokens/s without any)
    eng.profile_begin(every=PROFILE_EVERY)
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(i)
    barrier()
    dt = time.perf_counter() - t0
    gemm_seen = eng.profile_seen()
    gemm_ms, gemm_n, gemm_flops = eng.profile_end(dev)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        tokens = world * c["B"] * c["T"] * args.steps
        achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        line = {
            "metric": ("decoded tokens/sec at batch=100, beam=5, 36x2048 regions" + (", index-list region format" if indexed else ""))
                      if beam > 1 else "decoded tokens/sec, greedy, batch=100, 36x2048 regions",
            "value": tokens / dt, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s decode, batch 100 images/GPU, 36 regions x 2048-d, 10 slots, seq_len 20, vocab 10000 "
                                   "(BASELINE configs[%d])" % ("beam-5" if beam > 1 else "greedy", 2 if beam > 1 else 1),
                       "beam": beam, "batch_per_gpu": c["B"], "seq_len": c["T"], "parallelism": "images sharded, dp%d" % world},
            "roofline": {"bound": "mfma", "kernel": "gemm_nt_f32_kernel (v_mfma_f32_32x32x2_f32)", "achieved": achieved,
                         "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS,
                         "traffic": measured_traffic() if (beam > 1 and not indexed) else None, "launches": gemm_seen,
                         "launches_timed": gemm_n, "avg_launch_us": gemm_ms * 1e3 / max(gemm_n, 1),
                         "gemm_share_of_wall": gemm_ms * 1e-3 / max(gemm_n, 1) * gemm_seen / dt,
                         "algorithmic_flops_per_launch": gemm_flops / max(gemm_n, 1)},
        }
        if indexed:
            # the same data through the dense wire format of the reference (regions materialised once, outside the timing)
            dense = [(d, r.dense().contiguous()) for d, r in batches]
            with torch.no_grad():
                for i in range(2):
                    m.beam_search(dense[i & 1], [EOS, -1], beam, 1)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for i in range(args.steps):
                    m.beam_search(dense[i & 1], [EOS, -1], beam, 1)
                torch.cuda.synchronize(dev)
            line["config"]["dense_format_same_data_tokens_per_s"] = c["B"] * c["T"] * args.steps / (time.perf_counter() - t1)
            line["config"]["region_bytes_per_batch"] = {"index_lists": int(batches[0][1].slot_idx.numel() * 4),
                                                        "dense": int(dense[0][1].numel() * 4)}
        if world == 1 and not args.no_cpu and not indexed:
            line["cpu_baseline"] = cpu_baseline(weights, args.cpu_sample, beam)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
