"""Single-node multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm;
"gloo" in the CPU tests).  The reference has no distributed code at all; what is checked here is
"N ranks == 1 rank on the concatenated batch".

Decode shards IMAGES and needs no data-path collective (beams never cross images, CaptioningModel.py:107-109):
    lo, hi = shard_bounds(n_images, world, rank); ids = model.beam_search(...shard...); all = gather_ids(ids, n_images)
Training is data parallel: gradients are summed over ranks in flat buckets; both XE losses are normalised by GLOBAL
counts (coco_scripts/train.py:108-109: the word loss averages over B*(T-1) targets, the gate loss over the targets
that are not ignore_index = -1), so uneven shards (100 images on 8 GPUs = 13,13,13,13,12,12,12,12) and data-dependent
ignore counts give exactly the single-process loss and update.
"""
import torch
import torch.distributed as dist
import torch.nn.functional as F


def shard_bounds(n, world, rank):
    """contiguous shard [lo, hi) of n items: the first n % world ranks get one more."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_ids(local, n_total, group=None):
    """all-gather (b_local, ...) integer tensors of uneven first dimension into (n_total, ...)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    cap = max(shard_bounds(n_total, world, r)[1] - shard_bounds(n_total, world, r)[0] for r in range(world))
    pad = torch.zeros((cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    out = []
    for r, p in enumerate(parts):
        lo, hi = shard_bounds(n_total, world, r)
        out.append(p[:hi - lo])
    return torch.cat(out, 0)


def allreduce_gradients(params, bucket_bytes=64 << 20, group=None):
    """SUM-all-reduce .grad of params in flat buckets (few large collectives: xGMI rings are per-link bound, so
    bucket size matters more than count).  All buckets are launched asynchronously, then unpacked in order."""
    params = [p for p in params if p.grad is not None]
    buckets, cur, cur_bytes = [], [], 0
    for p in params:
        nb = p.grad.numel() * p.grad.element_size()
        if cur and cur_bytes + nb > bucket_bytes:
            buckets.append(cur)
            cur, cur_bytes = [], 0
        cur.append(p)
        cur_bytes += nb
    if cur:
        buckets.append(cur)
    pending = []
    for b in buckets:
        flat = torch.cat([p.grad.reshape(-1) for p in b])
        pending.append((b, flat, dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)))
    for b, flat, work in pending:
        work.wait()
        off = 0
        for p in b:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n


class DataParallelStep:
    """One optimisation step of the XE (train.py:99-113) or SCST (train.py:151-178) phase on this rank's shard.

    forward_fn(det, captions, ctrl_seq) -> (logp_words (b,T,V), logp_gates (b,T,2)) with a graph; on the GPU it is
    `lambda d, c, s: model((d,), (c, s))`.  sample_fn(det, ctrl) -> ((words, gates), (lp_w, lp_g))."""

    def __init__(self, params, optimizer, forward_fn=None, sample_fn=None, group=None, bucket_bytes=64 << 20):
        self.params = list(params)
        self.opt = optimizer
        self.forward_fn = forward_fn
        self.sample_fn = sample_fn
        self.group = group
        self.bucket_bytes = bucket_bytes

    def _world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def _sum(self, t):
        if self._world() > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def xe_step(self, det, captions, ctrl_seq, gate_gts):
        out, gate = self.forward_fn(det, captions, ctrl_seq)
        V = out.shape[-1]
        tgt_w = captions[:, 1:].reshape(-1)
        tgt_g = gate_gts.reshape(-1).long()
        # global denominators (no gradient flows through them)
        counts = torch.tensor([float(tgt_w.numel()), float((tgt_g != -1).sum())], dtype=torch.float64, device=out.device)
        self._sum(counts)
        nll_w = F.nll_loss(out[:, :-1].reshape(-1, V), tgt_w, reduction="sum")
        nll_g = F.nll_loss(gate.reshape(-1, 2), tgt_g, ignore_index=-1, reduction="sum")
        loss_cap = nll_w / counts[0].to(nll_w.dtype)
        loss_gate = nll_g / counts[1].to(nll_g.dtype)
        loss = loss_cap + 4 * loss_gate                      # this rank's share of the global loss
        self.opt.zero_grad()
        loss.backward()
        if self._world() > 1:
            allreduce_gradients(self.params, self.bucket_bytes, self.group)
        self.opt.step()
        stats = torch.stack([loss.detach(), loss_cap.detach(), loss_gate.detach()]).double()
        return self._sum(stats)                              # global loss, loss_cap, loss_gate

    def scst_step(self, det, ctrl, reward_fn):
        """reward_fn(words (b,T)) -> (reward (b,), baseline (b,)) tensors: the CIDEr side is the caller's (out of scope)."""
        (words, gates), (lp_w, lp_g) = self.sample_fn(det, ctrl)
        reward, baseline = reward_fn(words)
        n = torch.tensor([float(words.shape[0])], dtype=torch.float64, device=lp_w.device)
        self._sum(n)
        per = -(lp_w.mean(-1) + lp_g.mean(-1)) * (reward - baseline).to(lp_w.dtype)
        loss = per.sum() / n[0].to(per.dtype)
        self.opt.zero_grad()
        loss.backward()
        if self._world() > 1:
            allreduce_gradients(self.params, self.bucket_bytes, self.group)
        self.opt.step()
        return self._sum(loss.detach().double().reshape(1))[0]
