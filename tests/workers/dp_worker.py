"""One rank of the data-parallel GPU tests: DataParallelStep around the HIP ControllableCaptioningModel.

    python tests/workers/dp_worker.py --backend gloo|nccl --world N --rank R --port P --out file.npz [--same-gpu]

gloo + --same-gpu: every rank uses cuda:0 and the collectives go through host memory (runs on a 1-GPU box);
nccl: rank r uses cuda:r, RCCL collectives on the device (needs >= N GPUs)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "vsr-guided-cic_amd"), os.path.join(ROOT, "tests")]

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CFG = dict(V=300, B=5, R0=6, R=5, D=512, L=4, T=6, E=128, H=256, A=64)


def reward(words):
    r = (words.double().sum(1) % 7) / 7.0
    return r.float(), torch.full_like(r, 0.4).float()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--world", type=int, default=1)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--port", type=int, default=29611)
    ap.add_argument("--out", required=True)
    ap.add_argument("--same-gpu", action="store_true")
    a = ap.parse_args()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(a.port)
    dev = torch.device("cuda", 0 if a.same_gpu else a.rank)
    torch.cuda.set_device(dev)
    if a.world > 1:
        dist.init_process_group(a.backend, rank=a.rank, world_size=a.world)
    import helpers
    from vsrcap import parallel, synth
    cfg = CFG
    w = synth.make_weights(cfg["V"], cfg["D"], cfg["E"], cfg["H"], cfg["A"], seed=2, gains={k: 1.5 for k in synth.DEFAULT_GAINS})
    m = helpers.build_model(cfg, w, dev).train()
    # plain SGD: the update is linear in the gradient, so "N ranks == 1 rank" is tested on the exchange itself (Adam's
    # g / (|g| + eps) turns a 1e-9 rounding difference of a near-zero gradient into a 1e-4 difference of the weight)
    opt = torch.optim.SGD(m.parameters(), lr=0.05)
    det, seq, caps, gts = helpers.train_inputs(cfg, 9)
    d2, c2 = helpers.decode_inputs(cfg, 10)
    lo, hi = parallel.shard_bounds(cfg["B"], a.world, a.rank)

    host = None
    if a.world > 1 and a.backend == "gloo":
        def host(t):                      # SUM through host memory: gloo moves CPU tensors
            c = t.detach().cpu()
            dist.all_reduce(c)
            t.copy_(c)

    def sample_fn(d, c):
        with torch.no_grad():
            m.eval()
            w_, g_ = m.test(d, c)
            m.train()
        return m.sample_rl(d, c, forced=(w_, g_))

    step = parallel.DataParallelStep(m, opt, forward_fn=lambda d, c, s: m((d,), (c, s)), sample_fn=sample_fn, all_reduce_fn=host)
    sl = slice(lo, hi)
    losses = [step.xe_step(det[sl].to(dev), caps[sl].to(dev), seq[sl].to(dev), gts[sl].to(dev)).cpu().numpy() for _ in range(2)]
    rl = float(step.scst_step(d2[sl].to(dev), c2[sl].to(dev), reward))
    ids = None
    if a.world > 1:
        loc = torch.arange(lo, hi, device=dev)[:, None].repeat(1, 3)
        ids = (parallel.gather_ids(loc.cpu(), cfg["B"]) if a.backend == "gloo" else parallel.gather_ids(loc, cfg["B"]).cpu()).numpy()
    torch.cuda.synchronize(dev)
    if a.rank == 0:
        np.savez(a.out, losses=np.stack(losses), rl=np.array([rl]), ids=ids if ids is not None else np.zeros(0),
                 **{"p_" + k: v.detach().cpu().numpy() for k, v in m.named_parameters()})
    if a.world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
