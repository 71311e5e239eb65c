"""Build libvsrcap.so (hand-written HIP for gfx950) in-tree with hipcc.

    python vsr-guided-cic_amd/build.py [--force]

The shared library lands next to the ctypes binding (vsr-guided-cic_amd/vsrcap/libvsrcap.so) so that
it travels to the GPU box with the repo snapshot; it is git-ignored (source-only history).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "vsrcap.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "gemm_f32.h"), os.path.join(HERE, "csrc", "gemm_bf16.h"), os.path.join(HERE, "csrc", "gemm_f32x3.h"), os.path.join(HERE, "csrc", "ssp_kernels.h"), os.path.join(HERE, "csrc", "ssp.inc.h"), os.path.join(HERE, "csrc", "kernels.h"),
        os.path.join(HERE, "csrc", "train_kernels.h"), os.path.join(HERE, "csrc", "train.inc.h"),
        os.path.join(HERE, "csrc", "cider.inc.h"),
        os.path.join(os.path.dirname(HERE), "include", "vsrcap.h")]
OUT = os.path.join(HERE, "vsrcap", "libvsrcap.so")
# standalone GEMM check / timing tool (tools/README.md); its `fuzz` mode is run by tests/test_gpu_gemm_fuzz.py
TOOL_SRC = os.path.join(os.path.dirname(HERE), "tools", "gemm_bench.hip")
TOOL_DEPS = [TOOL_SRC, os.path.join(os.path.dirname(HERE), "tools", "gemm_bf16x3.h"), os.path.join(os.path.dirname(HERE), "tools", "gemm_dma_variant.h")] + DEPS[1:4]
TOOL_OUT = os.path.join(os.path.dirname(HERE), "tools", "gemm_bench")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-o", OUT + ".tmp", SRC]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    cmd[1:1] = os.environ.get("VSR_EXTRA_HIPCC_FLAGS", "").split()         # diagnostics builds (-DATT_ABLATE=..., -DGEMM_STAMP, ...)
    subprocess.run(cmd, check=True)
    os.replace(OUT + ".tmp", OUT)
    return OUT


def build_tool(force=False):
    if not force and os.path.exists(TOOL_OUT) and all(not os.path.exists(d) or os.path.getmtime(d) <= os.path.getmtime(TOOL_OUT) for d in TOOL_DEPS):
        return TOOL_OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-o", TOOL_OUT + ".tmp", TOOL_SRC], check=True)
    os.replace(TOOL_OUT + ".tmp", TOOL_OUT)
    return TOOL_OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
    if "--tool" in sys.argv:
        print(build_tool(force="--force" in sys.argv))
