"""The grouped GEMM kernels on their own, outside the decoder: tools/gemm_bench `fuzz` launches random ragged grouped problems
(1-3 problems of 1-3 k segments, row gathers, K tails, odd leading dimensions and offsets of the output window, 8-256 workgroup
slots, stream-K ranges and k-aligned pieces) and compares the slab sums with an fp64 host reference; columns outside the output
window must stay untouched.  Variants: the fp32 64x64 / 128x64 / 128x128 kernels, the rows-16 kernel, and the two 16-wave kernels
with hand-written asynchronous loads (bf16 with fp32 A and - 1665 - with bf16 images of A: tolerance of bf16 operands; f32x3 in both
tile widths, 128 x 256 and - "3300 21" - 128 x 128, and the weight-streaming f32x3 kernel of gemm_x3s.h - "3400 1" -: fp32 tolerance;
the f16x2 kernels of gemm_h2.h on fp16-pair weight images with measured scale exponents - "5200 1" / "5200 21" the 16-wave tiles,
"5300 1" the streaming kernel, "5400 1" / "5400 21" the all-DMA tiles of gemm_h2a.h whose A operands are images too -: fp32 tolerance)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
TOOL = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gemm_bench")


@pytest.mark.parametrize("variant", ["1 1", "121 1", "2 2", "1605 2", "1607 2", "1664 1", "1665 1", "1664 21", "1665 21", "3300 1", "3300 21",
                                     "3400 1",           # the weight-streaming f32x3 kernel (<= 128 rows, k-aligned pieces only)
                                     "5200 1", "5200 21", "5300 1", "5400 1", "5400 21",
                                     "5400 21 nw4",      # ... the 128 x 128 all-DMA tile with a DMA ring of four stages (the product's default for <= 128 rows)
                                     "1666 1", "1666 21"])
def test_gemm_variant_on_random_ragged_launches(variant):
    if not os.path.exists(TOOL):
        pytest.skip("tools/gemm_bench not built (python vsr-guided-cic_amd/build.py --tool, or __graft_entry__.build())")
    v = variant.split()
    env = dict(os.environ, H2_NW="4") if v[-1] == "nw4" else dict(os.environ)
    r = subprocess.run([TOOL, "fuzz"] + v[:2] + ["16", "5"], capture_output=True, text=True, timeout=600, env=env)
    tail = "\n".join(r.stdout.splitlines()[-20:])
    assert r.returncode == 0 and "0 of 16 cases failed" in r.stdout, tail + r.stderr[-2000:]
