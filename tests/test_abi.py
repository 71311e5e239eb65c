"""CPU checks of the drop-in boundary: the shared library builds/loads and exports exactly what the
header declares; the ctypes binding covers every symbol (no compute without a GPU)."""
import os
import re


from conftest import ROOT


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "vsrcap.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vsr_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from vsrcap import _lib
    import importlib.util
    spec = importlib.util.spec_from_file_location("vsr_build", os.path.join(ROOT, "vsr-guided-cic_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.build()
    lib = _lib.load()
    syms = _header_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), "libvsrcap.so does not export %s" % s
    assert sorted(_lib.SIGNATURES) == syms, "ctypes binding and include/vsrcap.h disagree"
    assert lib.vsr_abi_version() == 1


def test_struct_layouts_match_header():
    from vsrcap import _lib
    import ctypes
    assert ctypes.sizeof(_lib.VsrDims) == 9 * 4
    assert ctypes.sizeof(_lib.VsrWeights) == 28 * 8
    text = open(os.path.join(ROOT, "include", "vsrcap.h")).read()
    body = text[text.index("typedef struct vsr_weights {"):text.index("} vsr_weights;")]
    fields = re.findall(r"const float\*\s+(\w+);", body)
    assert fields == [f for f, _ in _lib.WEIGHT_FIELDS]


def test_create_rejects_bad_dims_or_reports_no_device():
    import ctypes as C
    from vsrcap import _lib
    lib = _lib.load()
    h = C.c_void_p()
    d = _lib.VsrDims(seq_len=20, vocab_size=100, bos_idx=2, det_feat_size=2046, input_encoding_size=1000, rnn_size=1000,
                     att_size=512, h2_first_lstm=1, img_second_lstm=0)
    assert lib.vsr_create(C.byref(d), C.byref(h)) != 0
    assert b"multiples of 4" in lib.vsr_last_error()
    assert lib.vsr_workspace_bytes(None, 1, 1, 1, 1, 1) == 0
