"""Single-node multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm;
"gloo" in the CPU tests).  The reference has no distributed code at all; what is checked here is
"N ranks == 1 rank on the concatenated batch".

Decode shards IMAGES and needs no data-path collective (beams never cross images, CaptioningModel.py:107-109):
    lo, hi = shard_bounds(n_images, world, rank); ids = model.beam_search(...shard...); all = gather_ids(ids, n_images)
Training is data parallel.  Both XE losses are normalised by GLOBAL counts (coco_scripts/train.py:108-109: the word loss
averages over B*(T-1) targets, the gate loss over the targets that are not ignore_index = -1), so uneven shards (100 images
on 8 GPUs = 13,13,13,13,12,12,12,12) and data-dependent ignore counts give exactly the single-process loss and update.

Gradient exchange: the 28 gradients live in ONE flat fp32 buffer (FlatGrads) laid out in the order in which
vsr_train_backward completes them (vsr_train_bucket_map: 5 buckets, largest first, 97 / 65 / 80 / 30 / 14 MB at the full
model); the parameters' .grad are views of it - no torch.cat, no copy back.  The library records a HIP event after each
bucket; the all-reduce of bucket b is launched on a side stream behind that event, so it runs under the weight-gradient
GEMMs of buckets b+1.. (xGMI rings are per-link bound: few, large, contiguous collectives).  The optimizer step waits for
the last bucket only.
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


class FlatGrads:
    """One flat fp32 gradient buffer; .views[i] is the gradient of params[i] and stays its .grad for the whole run.
    order / bucket_of: position of every parameter in the buffer (bucket-major) - contiguous ranges per bucket."""

    def __init__(self, params, bucket_of=None):
        self.params = list(params)
        n = len(self.params)
        bucket_of = list(bucket_of) if bucket_of is not None else [0] * n
        order = sorted(range(n), key=lambda i: (bucket_of[i], i))
        total = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        self.flat = torch.zeros(total, dtype=p0.dtype, device=p0.device)
        self.views = [None] * n
        self.ranges = []                     # per bucket: (lo, hi) in the flat buffer
        off, cur, lo = 0, None, 0
        for i in order:
            if bucket_of[i] != cur:
                if cur is not None:
                    self.ranges.append((lo, off))
                cur, lo = bucket_of[i], off
            k = self.params[i].numel()
            self.views[i] = self.flat[off:off + k].view_as(self.params[i])
            off += k
        self.ranges.append((lo, off))

    def attach(self):
        """(re)install the views as .grad (optimizer.zero_grad(set_to_none=True) would have dropped them)"""
        for p, v in zip(self.params, self.views):
            p.grad = v

    def bucket(self, b):
        lo, hi = self.ranges[b]
        return self.flat[lo:hi]


class DataParallelStep:
    """One optimisation step of the XE (train.py:99-113) or SCST (train.py:151-178) phase on this rank's shard.

    model_or_params: the HIP ControllableCaptioningModel (gradients then go straight from vsr_train_backward into the flat
    buffer and are exchanged bucket by bucket behind the library's events), or a plain parameter list (CPU tests with the
    oracle: autograd accumulates into the flat views in place, buckets are exchanged after backward).
    forward_fn(det, captions, ctrl_seq) -> (logp_words (b,T,V), logp_gates (b,T,2)) with a graph.
    sample_fn(det, ctrl) -> ((words, gates), (lp_w, lp_g)).
    all_reduce_fn(tensor) -> None: override of the SUM collective (tests: gloo through host memory).
    exchange_dtype: torch.float32 (default) or torch.bfloat16 - the wire format of the gradient exchange (BASELINE configs[3] names
    bf16: 142 MB instead of 285 MB per step over xGMI).  With bf16 every bucket is rounded to a bf16 image, the images are exchanged
    shard-wise (all-to-all), summed in fp32 by the shard's owner, rounded once more and all-gathered into the fp32 flat buffer the
    optimizer reads: gradients, Adam state and master weights stay fp32, an exchanged value is rounded twice whatever the world size."""

    def __init__(self, model_or_params, optimizer, forward_fn=None, sample_fn=None, group=None, all_reduce_fn=None,
                 exchange_dtype=torch.float32):
        self.model = model_or_params if hasattr(model_or_params, "_engine") else None
        self.opt = optimizer
        self.forward_fn = forward_fn
        self.sample_fn = sample_fn
        self.group = group
        self.all_reduce_fn = all_reduce_fn
        if exchange_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("exchange_dtype must be torch.float32 or torch.bfloat16")
        self.exchange_dtype = exchange_dtype
        self._wire = None                  # bf16 image of the flat gradient buffer (exchange_dtype = bf16, caller-supplied SUM)
        self._cap_key, self._cap = None, 0
        self._wires = {}                   # shared bf16 staging (send, recv, out) + fp32 shard of the all-to-all exchange, sized for the largest bucket
        if self.model is not None:
            from . import _lib
            sd = dict(self.model.named_parameters())
            self.params = [sd[k] for _, k in _lib.WEIGHT_FIELDS]
            dev = self.params[0].device
            self.eng = self.model._engine(dev)
            bucket_of, self.n_buckets = self.eng.bucket_map()
            self.grads = FlatGrads(self.params, bucket_of)
            self.eng.grad_sink = self.grads.views
            self.comm_stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        else:
            self.params = list(model_or_params)
            self.eng = None
            self.grads = FlatGrads(self.params)
            self.n_buckets = 1
            self.comm_stream = None

    def close(self):
        """Give the model back to plain autograd: while a DataParallelStep owns it, vsr_train_backward writes the gradients
        into this step's flat buffer (eng.grad_sink) and the autograd Function returns no tensors - a loss.backward() outside
        the step object would therefore leave p.grad untouched views of that buffer."""
        if self.eng is not None:
            self.eng.grad_sink = None
        for p in self.params:
            p.grad = None

    def _world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def _wire_cap(self, world):
        """elements of the shared bf16 staging buffers: the largest bucket, rounded up to whole shards (computed once per world size)"""
        if self._cap_key != (world, id(self.grads)):
            g = self.grads
            self._cap = max(world * ((hi - lo + world - 1) // world) for lo, hi in g.ranges)
            self._cap_key = (world, id(self.grads))
        return self._cap

    def _sum(self, t):
        if self._world() > 1:
            if self.all_reduce_fn is not None:
                self.all_reduce_fn(t)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def _exchange_bucket(self, buf):
        """SUM `buf` (a slice of the fp32 flat gradient buffer) over the ranks, in place, in the configured wire format.
        Returns the async work handle of the collective, or None when it has completed / was done by all_reduce_fn."""
        if self.exchange_dtype == torch.float32:
            if self.all_reduce_fn is not None:
                self.all_reduce_fn(buf)
                return None
            return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        g = self.grads
        if self.all_reduce_fn is not None:
            # caller-supplied SUM (tests that stage the collective through host memory): the bf16 image itself is summed
            if self._wire is None or self._wire.numel() != g.flat.numel() or self._wire.device != g.flat.device:
                self._wire = torch.empty(g.flat.numel(), dtype=torch.bfloat16, device=g.flat.device)
            lo = (buf.data_ptr() - g.flat.data_ptr()) // g.flat.element_size()
            wire = self._wire[lo:lo + buf.numel()]
            wire.copy_(buf)                                     # round to nearest even
            self.all_reduce_fn(wire)
            buf.copy_(wire)
            return None
        # bf16 on the wire, fp32 in the sum: every rank sends shard r of its bf16 image to rank r (all-to-all: on the fully connected
        # xGMI mesh every pair has its own link), rank r adds the `world` contributions in fp32 in rank order, rounds the sum to bf16
        # once, and the shards are all-gathered.  Same bytes per link as a ring all-reduce of the bf16 image (2 (N-1)/N x 142 MB),
        # but an element is rounded TWICE whatever the world size - a ring that sums in bf16 rounds it up to N times
        # (tests/test_parallel.py states the deviation at world 2 and 8).
        # Staging: ONE send / recv / out triple of bf16 and one fp32 shard, sized for the largest bucket and shared by all of them (the
        # buckets are exchanged one after the other on this stream): 1.5 x the largest bucket in bf16 + its shard in fp32 of resident
        # memory instead of a triple per bucket size, and no temporaries per call.
        world = self._world()
        n = buf.numel()
        shard = (n + world - 1) // world
        cap = self._wire_cap(world)
        if world * shard > cap:
            raise RuntimeError("bf16 exchange: a buffer of %d elements is larger than the largest gradient bucket (%d): "
                               "_exchange_bucket takes slices of this step's flat gradient buffer only" % (n, cap))
        key = (cap, world, buf.device)
        if self._wires.get("key") != key:
            self._wires = {"key": key,
                           "bf16": tuple(torch.zeros(cap, dtype=torch.bfloat16, device=buf.device) for _ in range(3)),
                           "f32": torch.zeros(cap // world, dtype=torch.float32, device=buf.device)}
        send, recv, out = (t[:world * shard] for t in self._wires["bf16"])
        red32 = self._wires["f32"][:shard]
        send[:n].copy_(buf)                                     # round to nearest even
        if world * shard > n:
            send[n:].zero_()                                    # (the padding of THIS bucket; the buffers are shared)
        dist.all_to_all_single(recv, send, group=self.group)    # stream-ordered on this (side) stream
        torch.sum(recv.view(world, shard), 0, dtype=torch.float32, out=red32)      # fp32 sum in rank order of the bf16 contributions
        red = send[:shard]                                      # (send is free again: the all-to-all is stream-ordered)
        red.copy_(red32)                                        # one rounding of the sum
        dist.all_gather_into_tensor(out, red, group=self.group)
        buf.copy_(out[:n])
        return None

    def _backward_and_exchange(self, loss):
        g = self.grads
        if self.model is not None and getattr(self.model, "_eng", None) is not self.eng:
            raise RuntimeError("the model's engine changed under this DataParallelStep (device move?): build a new one")
        if self.eng is None:
            g.flat.zero_()                   # autograd ACCUMULATES into existing .grad tensors: start from zero, in place
        g.attach()
        loss.backward()
        if self._world() == 1:
            return
        if self.eng is None or self.comm_stream is None:
            for b in range(len(g.ranges)):
                w = self._exchange_bucket(g.bucket(b))
                if w is not None:
                    w.wait()
            return
        # HIP path: the whole backward is enqueued (the host is ahead of the GPU); bucket b's collective goes to the side
        # stream behind the library's event for bucket b and overlaps the GEMMs of the later buckets
        main = torch.cuda.current_stream(g.flat.device)
        works = []
        for b in range(self.n_buckets):
            self.eng.wait_bucket(b, self.comm_stream)
            with torch.cuda.stream(self.comm_stream):
                w = self._exchange_bucket(g.bucket(b))
                if w is not None:
                    works.append(w)
        with torch.cuda.stream(self.comm_stream):
            for w in works:
                w.wait()                     # RCCL's internal stream -> side stream
        main.wait_stream(self.comm_stream)   # the optimizer step (main stream) needs every bucket

    def _optimizer_step(self):
        self.opt.step()
        if self.model is not None:
            # the weights moved: whatever the library derived from them (fp16-pair images, bf16 copies, decode cache) is redone by the
            # next call - said explicitly, not inferred from Tensor._version (fused optimizers leave it untouched)
            self.model.invalidate_cache()

    def xe_step(self, det, captions, ctrl_seq, gate_gts):
        out, gate = self.forward_fn(det, captions, ctrl_seq)
        V = out.shape[-1]
        tgt_w = captions[:, 1:].reshape(-1)
        tgt_g = gate_gts.reshape(-1).long()
        # global denominators (no gradient flows through them)
        counts = torch.tensor([float(tgt_w.numel()), 0.0], dtype=torch.float64, device=out.device)
        counts[1] = (tgt_g != -1).sum()
        self._sum(counts)
        nll_w = F.nll_loss(out[:, :-1].reshape(-1, V), tgt_w, reduction="sum")
        nll_g = F.nll_loss(gate.reshape(-1, 2), tgt_g, ignore_index=-1, reduction="sum")
        loss_cap = nll_w / counts[0].to(nll_w.dtype)
        loss_gate = nll_g / counts[1].to(nll_g.dtype)
        loss = loss_cap + 4 * loss_gate                      # this rank's share of the global loss
        self._backward_and_exchange(loss)
        self._optimizer_step()
        stats = torch.stack([loss.detach(), loss_cap.detach(), loss_gate.detach()]).double()
        return self._sum(stats)                              # global loss, loss_cap, loss_gate

    def scst_step(self, det, ctrl, reward_fn):
        """reward_fn(words (b,T)) -> (reward (b,), baseline (b,)) tensors (vsrcap.reward.CiderD on the device, or the caller's)."""
        (words, gates), (lp_w, lp_g) = self.sample_fn(det, ctrl)
        reward, baseline = reward_fn(words)
        n = torch.tensor([float(words.shape[0])], dtype=torch.float64, device=lp_w.device)
        self._sum(n)
        per = -(lp_w.mean(-1) + lp_g.mean(-1)) * (reward - baseline).to(lp_w.dtype)
        loss = per.sum() / n[0].to(per.dtype)
        self._backward_and_exchange(loss)
        self._optimizer_step()
        return self._sum(loss.detach().double().reshape(1))[0]
