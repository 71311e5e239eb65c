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
GEMM_HEADERS = sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc")) if f.startswith("gemm_"))
DEPS = sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc")) if f.endswith((".h", ".hip"))) + \
       [os.path.join(os.path.dirname(HERE), "include", "vsrcap.h")]
OUT = os.path.join(HERE, "vsrcap", "libvsrcap.so")
# standalone GEMM check / timing tool (tools/README.md); its `fuzz` mode is run by tests/test_gpu_gemm_fuzz.py
TOOL_SRC = os.path.join(os.path.dirname(HERE), "tools", "gemm_bench.hip")
TOOL_DEPS = [TOOL_SRC] + GEMM_HEADERS
TOOL_OUT = os.path.join(os.path.dirname(HERE), "tools", "gemm_bench")


STAMP = OUT + ".flags"      # the extra hipcc flags the library on disk was built with


def _flags():
    return os.environ.get("VSR_EXTRA_HIPCC_FLAGS", "").strip()


def needs_build():
    if not os.path.exists(OUT):
        return True
    # a library built with other extra flags (a diagnostics build) must never be picked up by a later plain run.  A shipped
    # library without a stamp file (the GPU box receives the .so, the stamp travels with it when present) counts as plain.
    built_with = open(STAMP).read().strip() if os.path.exists(STAMP) else ""
    if built_with != _flags():
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
    cmd[1:1] = _flags().split()              # experiments only (e.g. -save-temps); recorded in the stamp so that a later plain run rebuilds
    subprocess.run(cmd, check=True)
    os.replace(OUT + ".tmp", OUT)
    with open(STAMP, "w") as f:
        f.write(_flags())
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
