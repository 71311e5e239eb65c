import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "vsr-guided-cic_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# ---- every fp32 GEMM flavour in ONE pytest invocation -------------------------------------------------------------------
# Every `-m gpu` test of the decoder modules below runs three times: with the default flavour 'f16x2' (two fp16 terms per fp32
# operand under a power-of-two scale, three MFMAs per product, fp16-pair weight images: csrc/gemm_h2.h), with 'f32x3' (three bf16
# terms, six MFMAs: csrc/gemm_x3.h / gemm_x3s.h) and with 'f32' (the exact fma chain for every launch).  Same fixtures, same
# bounds: that is what admits a flavour as the parity-mode default.  The flavour is applied through
# models.set_default_compute_dtype(), i.e. the compute dtype a freshly constructed model starts in; tests that pick a dtype
# themselves (test_gpu_bf16.py, test_gpu_f32x3.py, test_gpu_h2.py) are not multiplied.  Child processes (the data-parallel
# workers, bench.py) receive it as VSR_COMPUTE_DTYPE.
FLAVOURS = ("f16x2", "f32x3", "f32")
DUAL_FLAVOUR_MODULES = ("test_gpu_parity", "test_gpu_configs", "test_gpu_train", "test_gpu_regions", "test_aa_gpu_dp",
                        "test_gpu_headline", "test_gpu_train_indexed", "test_gpu_real_shapes", "test_gpu_weight_generation",
                        "test_gpu_traj", "test_gpu_live_forwards")


def pytest_generate_tests(metafunc):
    mod = metafunc.module.__name__.rsplit(".", 1)[-1]
    if "gemm_flavour" in metafunc.fixturenames and mod in DUAL_FLAVOUR_MODULES and metafunc.definition.get_closest_marker("gpu"):
        metafunc.parametrize("gemm_flavour", FLAVOURS, indirect=True, scope="session")


@pytest.fixture(scope="session", autouse=True)
def gemm_flavour(request):
    flavour = getattr(request, "param", None)
    if flavour is None:
        yield None
        return
    import models
    old = models.set_default_compute_dtype(flavour)
    old_env = os.environ.get("VSR_COMPUTE_DTYPE")
    os.environ["VSR_COMPUTE_DTYPE"] = flavour
    yield flavour
    models.set_default_compute_dtype(old)
    if old_env is None:
        os.environ.pop("VSR_COMPUTE_DTYPE", None)
    else:
        os.environ["VSR_COMPUTE_DTYPE"] = old_env


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    return meta, {k: z[k] for k in z.files if k != "meta"}


@pytest.fixture(scope="session")
def golden():
    return load_golden
