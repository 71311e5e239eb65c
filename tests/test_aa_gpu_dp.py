"""Data-parallel step around the HIP model on real hardware: N ranks == 1 rank on the concatenated batch (SURVEY 8e check:
loss <= 1e-4, updated weights <= 1e-6 relative), with UNEVEN shards (5 images on 2 ranks) and data-dependent ignore counts.

The file name sorts first on purpose: the ranks are child processes, and this (parent) process must not have initialised
the GPU before it starts them (torch.cuda.device_count() does not).  Every rank, also the 1-rank baseline, is a child.
  * gloo, both ranks on cuda:0, collectives staged through host memory   -> runs on the 1-GPU box
  * nccl (= RCCL), rank r on cuda:r                                      -> skipped below 2 GPUs
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "workers", "dp_worker.py")


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(tmp_path, backend, world, same_gpu, tag):
    out = str(tmp_path / ("%s.npz" % tag))
    port = _port()
    procs = []
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for r in range(world):
        cmd = [sys.executable, WORKER, "--backend", backend, "--world", str(world), "--rank", str(r), "--port", str(port), "--out", out]
        if same_gpu:
            cmd.append("--same-gpu")
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    return np.load(out)


def _compare(one, two):
    np.testing.assert_allclose(two["losses"], one["losses"], atol=1e-4, rtol=0)          # loss, loss_cap, loss_gate per step
    assert abs(float(two["rl"][0]) - float(one["rl"][0])) < 1e-5
    for k in one.files:
        if k.startswith("p_"):
            a, b = one[k].astype(np.float64), two[k].astype(np.float64)
            err = float(np.abs(a - b).max())
            assert err <= 1e-6 * max(float(np.abs(a).max()), 1e-3) + 1e-6, "%s: max |dw| %.3e" % (k, err)      # 1e-6 relative
    np.testing.assert_array_equal(two["ids"][:, 0], np.arange(5))


def test_two_ranks_one_gpu_host_staged_exchange(tmp_path):
    one = _run(tmp_path, "gloo", 1, True, "one")
    two = _run(tmp_path, "gloo", 2, True, "two")
    _compare(one, two)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs (RCCL over xGMI)")
def test_two_ranks_two_gpus_rccl(tmp_path):
    one = _run(tmp_path, "nccl", 1, False, "one")
    two = _run(tmp_path, "nccl", 2, False, "two")
    _compare(one, two)


def test_rccl_collectives_of_the_exchange_on_one_rank(tmp_path):
    """The data-parallel exchange's RCCL calls on real hardware, as far as a 1-GPU box allows: a ONE-rank "nccl" (= RCCL) process group on
    cuda:0, then the two wire formats of DataParallelStep._exchange_bucket on a plain parameter list - fp32 all-reduce, and the bf16 wire
    (all-to-all of the bf16 image, fp32 sum, all-gather) through the shared staging buffers - bucket by bucket.  With one rank the sum
    is the rank's own contribution: fp32 comes back bit-identical, bf16 comes back rounded to bf16 once (a second rounding of a bf16
    value changes nothing).  (More than one rank: test_two_ranks_two_gpus_rccl, skipped below 2 GPUs.)"""
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(%r, "vsr-guided-cic_amd"))
from vsrcap import parallel
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ["MASTER_PORT"] = "%d"
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
params = [torch.nn.Parameter(torch.randn(n, device="cuda")) for n in (1000, 37, 4096, 513)]
opt = torch.optim.SGD(params, lr=0.1)
for dt in (torch.float32, torch.bfloat16):
    st = parallel.DataParallelStep(params, opt, exchange_dtype=dt)
    st.grads = parallel.FlatGrads(params, bucket_of=[0, 0, 1, 2])
    g = st.grads
    g.flat.copy_(torch.randn_like(g.flat) * 3.0)
    want = g.flat.clone() if dt == torch.float32 else g.flat.to(torch.bfloat16).float()
    for b in range(len(g.ranges)):
        w = st._exchange_bucket(g.bucket(b))
        if w is not None:
            w.wait()
    torch.cuda.synchronize()
    assert torch.equal(g.flat, want), (dt, (g.flat - want).abs().max().item())
t = torch.ones(8, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
assert float(t.sum()) == 8.0
dist.destroy_process_group()
print("rccl one-rank exchange ok")
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), _port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl one-rank exchange ok" in r.stdout, (r.stdout + r.stderr)[-3000:]
