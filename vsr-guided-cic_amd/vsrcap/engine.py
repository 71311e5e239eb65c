"""Thin host-side driver of the C ABI: owns one vsr_handle, borrows torch storage for weights / tensors /
workspace and enqueues everything on torch's current HIP stream.  PyTorch is plumbing here (device memory,
streams); all arithmetic happens in libvsrcap.so.
"""
import ctypes as C
import functools
import os

import numpy as np
import torch

from . import _lib


def _on_device(fn):
    """Run an Engine method with the handle's device current: launches go to torch.cuda.current_stream(dev), and HIP
    wants that stream's device to be the current one (a model on cuda:1 used without torch.cuda.set_device(1))."""
    @functools.wraps(fn)
    def wrapped(self, *a, **kw):
        if self.device is None:
            return fn(self, *a, **kw)
        with torch.cuda.device(self.device):
            return fn(self, *a, **kw)
    return wrapped


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _f32(t, name):
    if t.dtype != torch.float32:
        t = t.float()
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU (got %s); this path has no CPU implementation" % (name, t.device))
    return t.contiguous()


_UNVERSIONED = [0]


def _ver(t):
    """in-place-write counter of a tensor for the prepare() cache key.  Tensors created under torch.inference_mode() track none
    (reading t._version raises): such an input gets a fresh key every time, i.e. it is never served from the cache."""
    if t is None:
        return None
    try:
        return t._version
    except RuntimeError:
        _UNVERSIONED[0] += 1
        return ("unversioned", _UNVERSIONED[0])


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class ForwardToken:
    """Lives in the autograd node of one training forward (train._DecoderFn): while it is alive and its forward has not been
    differentiated, the buffers that forward was saved in are not handed to anyone else."""
    __slots__ = ("generation", "__weakref__")

    def __init__(self, generation):
        self.generation = generation


class _Slot:
    """One pair of caller buffers of the C ABI - the vsr_prepare*() workspace and the training workspace - plus the inputs they borrow.
    A saved training forward lives in exactly one slot (include/vsrcap.h, vsr_train_select)."""
    __slots__ = ("ws", "tws", "keep", "token", "generation", "differentiated")

    def __init__(self):
        self.ws = self.tws = self.keep = self.token = None
        self.generation = 0
        self.differentiated = True

    def live(self):
        """holds a forward somebody can still ask the gradient of for the FIRST time"""
        return self.token is not None and self.token() is not None and not self.differentiated


MAX_LIVE_FORWARDS = 8


class Engine:
    def __init__(self, dims, device=None):
        """dims: dict with the vsr_dims fields; device: the torch cuda device this handle lives on (one handle per device)."""
        self.lib = _lib.load()
        self.dims = _lib.VsrDims(**dims)
        self.h = C.c_void_p()
        self.device = torch.device(device) if device is not None else None
        if self.device is not None and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        with torch.cuda.device(self.device) if self.device is not None else _Null():
            _lib.check(self.lib.vsr_create(C.byref(self.dims), C.byref(self.h)))
        # VSR_CHECK_IDS=1 (or eng.check_ids = True): after every call that takes word / slot / verb ids, read back the
        # library's count of out-of-range ids and raise like nn.Embedding would (costs one stream synchronisation per call)
        self.check_ids = os.environ.get("VSR_CHECK_IDS", "0") not in ("", "0")
        self._bound_ptrs = None
        self._slots = [_Slot()]          # buffer pairs; more than one only while several training forwards are alive
        self._slot = self._slots[0]
        self._prep_key = None
        self._verb_dev = None
        self.grad_sink = None       # list of 28 tensors (WEIGHT_FIELDS order): backward writes there and autograd gets no tensors

    # the current slot's buffers under their old names
    _ws = property(lambda self: self._slot.ws, lambda self, v: setattr(self._slot, "ws", v))
    _tws = property(lambda self: self._slot.tws, lambda self, v: setattr(self._slot, "tws", v))
    _keep = property(lambda self: self._slot.keep, lambda self, v: setattr(self._slot, "keep", v))

    def _claim_slot(self):
        """Called before a vsr_prepare*() that will overwrite the current slot's workspace: if a live forward is saved there, move to
        a free slot (one whose forward is gone or already differentiated), or open a new one.  The reference has no such limit at
        all (eager autograd); MAX_LIVE_FORWARDS pairs of buffers is where this build stops."""
        if not self._slot.live():
            return
        for sl in self._slots:
            if not sl.live():
                self._slot = sl
                self._prep_key = None
                return
        if len(self._slots) >= MAX_LIVE_FORWARDS:
            raise RuntimeError("%d training forwards of this model are alive and not yet differentiated: each holds its own workspaces "
                               "(~GBs at batch 100). Call backward() on some of them (or drop their outputs) before the next forward."
                               % len(self._slots))
        self._slot = _Slot()
        self._slots.append(self._slot)
        self._prep_key = None

    def live_forwards(self):
        return sum(1 for sl in self._slots if sl.live())

    def __del__(self):
        try:
            if self.h:
                self.lib.vsr_destroy(self.h)
                self.h = C.c_void_p()
        except Exception:
            pass

    # ------------------------------------------------------------------ weights / tables
    def bind(self, params):
        """params: dict state_dict key -> fp32 CUDA tensor (borrowed, not copied)."""
        ptrs = []
        for _, key in _lib.WEIGHT_FIELDS:
            t = params[key]
            if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
                raise RuntimeError("weight %s must be a contiguous fp32 GPU tensor (got %s %s)" % (key, t.dtype, t.device))
            ptrs.append(t.data_ptr())
        ptrs = tuple(ptrs)
        if ptrs != self._bound_ptrs:
            w = _lib.VsrWeights(*ptrs)
            _lib.check(self.lib.vsr_bind_weights(self.h, C.byref(w)))
            self._bound_ptrs = ptrs
            self._prep_key = None
            self._cache_key = None
        return ptrs

    def invalidate(self):
        """Forget the cached prepare() / decode cache.  prepare() keys its cache on (data_ptr, tensor._version, shapes,
        weights generation): writes that do not bump _version (t.data.copy_, DLPack / custom-kernel writes into the same
        buffer, p.data edits of the weights) are invisible to it - call this after such a write."""
        self._prep_key = None
        self._cache_key = None
        if getattr(self, "_bf16_key", None) is not None:
            self._bf16_key = ("stale",)          # bf16 mode stays on; the copies are rebuilt by the next set_bf16()
        if getattr(self, "_h2_key", None) is not None:
            self._h2_key = ("stale",)            # ... and so are the fp16-pair images

    def raise_on_bad_ids(self, device, who):
        if not self.check_ids:
            return
        n = C.c_int32(0)
        _lib.check(self.lib.vsr_bad_ids(self.h, C.byref(n), self._stream(device)))
        if n.value:
            raise IndexError("%s: %d word / slot / verb ids out of range (clamped on the device; nn.Embedding would raise) or region rows beyond "
                             "the caller's set_valid_rows_bound() (they got no att_va projection)" % (who, n.value))

    @_on_device
    def set_verb_table(self, table, device):
        """table: dict str(verb id) -> list of vocab ids (verb_2_vob_all of the reference)."""
        ids = [int(k) for k in table.keys()]
        n = (max(ids) + 1) if ids else 0
        row_ptr = np.zeros(n + 1, dtype=np.int32)
        flat = []
        for v in range(n):
            lst = table.get(str(v), [])
            flat.extend(int(x) for x in lst)
            row_ptr[v + 1] = len(flat)
        rp = torch.from_numpy(row_ptr).to(device)
        fl = torch.tensor(flat if flat else [0], dtype=torch.int32, device=device)
        self._verb_dev = (rp, fl)
        _lib.check(self.lib.vsr_set_verb_table(self.h, _ptr(rp), _ptr(fl), n))

    def set_gemm_mode(self, x3):
        """True (the library's default): "f32x3" - three bf16 terms per fp32 operand, six MFMAs per product; which kernel a launch
        takes depends on its rows (include/vsrcap.h, vsr_set_gemm_mode); False: the exact fp32 fma chain for every launch"""
        x3 = bool(x3)
        if x3 != getattr(self, "_x3", True):
            _lib.check(self.lib.vsr_set_gemm_mode(self.h, 1 if x3 else 0))
            self._x3 = x3
            self._cache_key = None
            self._prep_key = None

    # ------------------------------------------------------------------ f16x2: fp16-pair weight images on top of f32x3
    @_on_device
    def set_h2(self, device, weights_version, enable):
        """enable: (re)build the fp16-pair images of the weight matrices when the bound weights or their version changed
        (include/vsrcap.h, vsr_refresh_h2_weights); disable: back to plain f32x3."""
        key = (self._bound_ptrs, weights_version) if enable else None
        if key == getattr(self, "_h2_key", None):
            return
        if not enable:
            _lib.check(self.lib.vsr_refresh_h2_weights(self.h, C.c_void_p(0), 0, self._stream(device)))
        else:
            n = self.lib.vsr_h2_weight_bytes(self.h)
            if getattr(self, "_h2_buf", None) is None or self._h2_buf.numel() < n or self._h2_buf.device != device:
                self._h2_buf = torch.empty(n, dtype=torch.uint8, device=device)      # (torch allocations are 512-byte aligned)
            _lib.check(self.lib.vsr_refresh_h2_weights(self.h, _ptr(self._h2_buf), self._h2_buf.numel(), self._stream(device)))
        self._cache_key = None                   # the library voids the decode cache and the hoisted tensors on every refresh
        self._prep_key = None
        self._h2_key = key

    # ------------------------------------------------------------------ bf16 throughput mode
    @_on_device
    def set_bf16(self, device, weights_version, enable):
        """enable: (re)build the bf16 copies of the weight matrices when the bound weights or their version changed and
        switch the handle to bf16 GEMMs; disable: back to the fp32 parity mode (include/vsrcap.h, vsr_refresh_bf16_weights)."""
        key = (self._bound_ptrs, weights_version) if enable else None
        if key == getattr(self, "_bf16_key", None):
            return
        if not enable:
            _lib.check(self.lib.vsr_refresh_bf16_weights(self.h, C.c_void_p(0), 0, self._stream(device)))
        else:
            n = self.lib.vsr_bf16_weight_bytes(self.h)
            if getattr(self, "_bf16_buf", None) is None or self._bf16_buf.numel() < n or self._bf16_buf.device != device:
                self._bf16_buf = torch.empty(n, dtype=torch.uint8, device=device)
            _lib.check(self.lib.vsr_refresh_bf16_weights(self.h, _ptr(self._bf16_buf), self._bf16_buf.numel(), self._stream(device)))
        if (key is None) != (getattr(self, "_bf16_key", None) is None):
            self._cache_key = None               # the decode cache was built in the other precision
            self._prep_key = None                # ... and so were the hoisted projections
        self._bf16_key = key

    # ------------------------------------------------------------------ decode cache (inference only)
    @_on_device
    def decode_cache(self, device, weights_version, enable=True):
        """(Re)build the embedding-projection cache when the weights changed; enable=False drops it (training)."""
        key = (self._bound_ptrs, weights_version) if enable else None
        if key == getattr(self, "_cache_key", None):
            return
        if not enable:
            _lib.check(self.lib.vsr_build_decode_cache(self.h, C.c_void_p(0), 0, self._stream(device)))
            self._cache_key = None
            return
        n = self.lib.vsr_decode_cache_floats(self.h)
        if getattr(self, "_cache_buf", None) is None or self._cache_buf.numel() < n or self._cache_buf.device != device:
            self._cache_buf = torch.empty(n, dtype=torch.float32, device=device)
        _lib.check(self.lib.vsr_build_decode_cache(self.h, _ptr(self._cache_buf), self._cache_buf.numel(), self._stream(device)))
        self._cache_key = key

    # ------------------------------------------------------------------ hoisted statics
    @_on_device
    def prepare(self, det, regions, beam, weights_version=None, rows_bound=None, for_training=False):
        """rows_bound: an upper bound on the non-padding region rows the caller knows on the host (None: the library reads the
        count back - its one synchronisation; include/vsrcap.h, vsr_set_valid_rows_bound).
        for_training: a training forward follows (it overwrites the slot's training workspace): never served from the cache of a slot
        that holds a live forward"""
        det = _f32(det, "detections")
        regions = _f32(regions, "region sequences")
        if det.dim() != 3 or regions.dim() != 4 or det.size(0) != regions.size(0) or det.size(2) != regions.size(3):
            raise RuntimeError("expected detections (B,R0,D) and regions (B,L,R,D); got %s and %s" % (tuple(det.shape), tuple(regions.shape)))
        if det.size(2) != self.dims.det_feat_size:
            raise RuntimeError("feature size %d != det_feat_size %d" % (det.size(2), self.dims.det_feat_size))
        B, R0, _ = det.shape
        _, L, R, _ = regions.shape
        key = (det.data_ptr(), _ver(det), regions.data_ptr(), _ver(regions), B, R0, L, R, beam,
               self._bound_ptrs, weights_version, rows_bound)
        if key == self._prep_key and not (for_training and self._slot.live()):
            return B
        self._claim_slot()
        _lib.check(self.lib.vsr_set_valid_rows_bound(self.h, int(rows_bound or 0)))
        need = self.lib.vsr_workspace_bytes(self.h, B, R0, L, R, beam)
        if need == 0:
            raise RuntimeError("vsr_workspace_bytes rejected the shapes")
        if self._ws is None or self._ws.numel() < need or self._ws.device != det.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=det.device)
        stream = torch.cuda.current_stream(det.device).cuda_stream
        _lib.check(self.lib.vsr_prepare(self.h, _ptr(det), B, R0, _ptr(regions), L, R, beam, _ptr(self._ws),
                                        self._ws.numel(), C.c_void_p(stream)))
        self._keep = (det, regions)          # borrowed by the library until the next prepare
        self._prep_key = key
        return B

    @_on_device
    def prepare_indexed(self, det, bank, slot_idx, row_img, beam, weights_version=None, rows_bound=None, for_training=False):
        """Index-list region format (include/vsrcap.h, vsr_prepare_indexed): det (n_img,R0,D), bank (n_img,Rb,D),
        slot_idx (B,L,R) int32 rows of the row's image bank (-1 = padding), row_img (B) int32 or None."""
        det = _f32(det, "detections")
        bank = _f32(bank, "feature bank")
        if det.dim() != 3 or bank.dim() != 3 or slot_idx.dim() != 3 or det.size(0) != bank.size(0) or det.size(2) != bank.size(2):
            raise RuntimeError("expected detections (n_img,R0,D), bank (n_img,Rb,D), slot_idx (B,L,R); got %s, %s, %s"
                               % (tuple(det.shape), tuple(bank.shape), tuple(slot_idx.shape)))
        if det.size(2) != self.dims.det_feat_size:
            raise RuntimeError("feature size %d != det_feat_size %d" % (det.size(2), self.dims.det_feat_size))
        if not slot_idx.is_cuda or slot_idx.dtype != torch.int32:
            raise RuntimeError("slot_idx must be an int32 GPU tensor (got %s on %s)" % (slot_idx.dtype, slot_idx.device))
        slot_idx = slot_idx.contiguous()
        n_img, R0, _ = det.shape
        Rb = bank.size(1)
        B, L, R = slot_idx.shape
        if row_img is None:
            if B != n_img:
                raise RuntimeError("slot_idx has %d rows for %d images: pass row_img" % (B, n_img))
        else:
            if not row_img.is_cuda or row_img.dtype != torch.int32 or row_img.numel() != B:
                raise RuntimeError("row_img must be an int32 GPU tensor with one entry per row of slot_idx")
            row_img = row_img.contiguous()
        key = ("idx", det.data_ptr(), _ver(det), bank.data_ptr(), _ver(bank), slot_idx.data_ptr(), _ver(slot_idx),
               None if row_img is None else (row_img.data_ptr(), _ver(row_img)), B, n_img, R0, Rb, L, R, beam,
               self._bound_ptrs, weights_version, rows_bound)
        if key == self._prep_key and not (for_training and self._slot.live()):
            return B
        self._claim_slot()
        _lib.check(self.lib.vsr_set_valid_rows_bound(self.h, int(rows_bound or 0)))
        need = self.lib.vsr_workspace_bytes_indexed(self.h, B, R0, n_img, Rb, L, R, beam)
        if need == 0:
            raise RuntimeError("vsr_workspace_bytes_indexed rejected the shapes")
        if self._ws is None or self._ws.numel() < need or self._ws.device != det.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=det.device)
        self._prep_key = None
        _lib.check(self.lib.vsr_prepare_indexed(self.h, _ptr(det), n_img, R0, _ptr(bank), Rb, _ptr(row_img), B, _ptr(slot_idx),
                                                L, R, beam, _ptr(self._ws), self._ws.numel(), self._stream(det.device)))
        self._keep = (det, bank, slot_idx, row_img)      # borrowed by the library until the next prepare
        self._prep_key = key
        return B

    @_on_device
    def row_mask(self, rows):
        """(n, D) fp32 GPU rows -> (n,) fp32 mask of rows whose sum is not zero (the reference's zero-row test)."""
        rows = _f32(rows, "rows")
        flat = rows.reshape(-1, rows.size(-1))
        out = torch.empty(flat.size(0), dtype=torch.float32, device=rows.device)
        _lib.check(self.lib.vsr_row_mask(_ptr(flat), flat.size(0), flat.size(1), _ptr(out), self._stream(rows.device)))
        return out.reshape(rows.shape[:-1])

    @_on_device
    def reorder_slots(self, slot_idx, rank, verbs, bank_mask, row_img, Rb):
        """eval_coco.py:222-241 on index lists (vsr_reorder_slots): returns (slot_idx_out, verbs_out or None)."""
        N, L, R = slot_idx.shape
        out = torch.empty_like(slot_idx)
        vout = torch.empty(N, L, dtype=torch.float32, device=slot_idx.device) if verbs is not None else None
        _lib.check(self.lib.vsr_reorder_slots(_ptr(slot_idx), _ptr(rank), _ptr(verbs), _ptr(bank_mask), _ptr(row_img), N, L, R, Rb,
                                              _ptr(out), _ptr(vout), self._stream(slot_idx.device)))
        return out, vout

    def _stream(self, dev):
        return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    # ------------------------------------------------------------------ training (forward with saves + BPTT backward)
    @_on_device
    def train_forward(self, B, device, word_in, slots=None):
        T = word_in.size(1)
        V = self.dims.vocab_size
        word_in = word_in.to(device=device, dtype=torch.int64).contiguous()
        if slots is not None:
            slots = slots.to(device=device, dtype=torch.int64).contiguous()
        need = self.lib.vsr_train_workspace_bytes(self.h, B, T)
        if need == 0:
            raise RuntimeError("vsr_train_workspace_bytes rejected the shapes (prepare() first, same batch)")
        if getattr(self, "_tws", None) is None or self._tws.numel() < need or self._tws.device != device:
            self._tws = torch.empty(need, dtype=torch.uint8, device=device)
        out = torch.empty(B, T, V, dtype=torch.float32, device=device)
        gate = torch.empty(B, T, 2, dtype=torch.float32, device=device)
        if self._slot.live():
            raise RuntimeError("internal: training forward into a slot that holds a live forward (prepare(for_training=True) first)")
        _lib.check(self.lib.vsr_train_forward(self.h, _ptr(word_in), _ptr(slots), T, _ptr(out), _ptr(gate), _ptr(self._tws),
                                              self._tws.numel(), self._stream(device)))
        self._slot.generation = self.train_generation()
        self._slot.token = None
        self._slot.differentiated = True      # (until train.py attaches the autograd node's token: note_forward)
        self.raise_on_bad_ids(device, "train_forward")
        return out, gate

    def note_forward(self):
        """the autograd node of the forward just taken: returns the token that keeps its buffers reserved"""
        import weakref
        tok = ForwardToken(self._slot.generation)
        self._slot.token = weakref.ref(tok)
        self._slot.differentiated = False
        return tok

    def train_generation(self):
        """identity of the forward pass saved in the handle (0 = none): see vsr_train_generation in include/vsrcap.h"""
        return int(self.lib.vsr_train_generation(self.h))

    @_on_device
    def train_backward(self, device, grad_out, grad_gate, shapes, generation=None, into=None):
        """shapes: list of the 28 parameter shapes in WEIGHT_FIELDS order -> list of gradient tensors.
        generation: train_generation() recorded right after the forward this backward belongs to.
        into: optional list of 28 caller tensors (e.g. views of ONE flat buffer, parallel.FlatGrads) the library writes
        the gradients to instead of fresh allocations."""
        if generation is not None:
            # several forwards may be alive (each in its own slot): make this one the handle's current forward again
            if self.lib.vsr_train_select(self.h, int(generation), self._stream(device)) != 0:
                raise RuntimeError(
                    "backward of a forward pass whose saved activations are gone (%s). A forward stays differentiable until its "
                    "buffers are reused: that happens once it HAS been differentiated (a second backward needs a new forward - "
                    "retain_graph=True is not supported across a later forward), when more than %d forwards are alive, or when "
                    "the compute dtype / the parameters' storage changed in between."
                    % (self.lib.vsr_last_error().decode(), MAX_LIVE_FORWARDS))
            if self._slot.generation != generation:
                for sl in self._slots:
                    if sl.generation == generation:
                        self._slot = sl
                        self._prep_key = None           # the handle's hoisted state is this slot's now
                        break
            self._slot.differentiated = True
        grad_out = _f32(grad_out, "grad of word log-probs")
        grad_gate = _f32(grad_gate, "grad of gate log-probs")
        if into is not None:
            grads = list(into)
            for g, sh in zip(grads, shapes):
                if tuple(g.shape) != tuple(sh) or g.dtype != torch.float32 or not g.is_cuda or not g.is_contiguous():
                    raise RuntimeError("gradient sink tensor does not match its parameter (%s vs %s)" % (tuple(g.shape), tuple(sh)))
        else:
            grads = [torch.empty(s, dtype=torch.float32, device=device) for s in shapes]
        g = _lib.VsrWeights(*[t.data_ptr() for t in grads])
        _lib.check(self.lib.vsr_train_backward(self.h, _ptr(grad_out), _ptr(grad_gate), C.byref(g), self._stream(device)))
        return grads

    def bucket_map(self):
        """(bucket_of[28] in WEIGHT_FIELDS order, n_buckets): completion order of the gradients in vsr_train_backward"""
        arr = (C.c_int32 * 28)()
        n = C.c_int32(0)
        _lib.check(self.lib.vsr_train_bucket_map(arr, C.byref(n)))
        return list(arr), n.value

    @_on_device
    def wait_bucket(self, bucket, stream):
        """make torch stream `stream` wait on the device until gradient bucket `bucket` of the last backward is complete"""
        _lib.check(self.lib.vsr_train_wait_bucket(self.h, int(bucket), C.c_void_p(stream.cuda_stream)))

    @_on_device
    def debug_buffer(self, name, shape, device):
        out = torch.empty(shape, dtype=torch.float32, device=device)
        _lib.check(self.lib.vsr_debug_copy(self.h, name.encode(), _ptr(out), out.numel(), self._stream(device)))
        return out

    # ------------------------------------------------------------------ measurement
    def profile_begin(self, every=1):
        """HIP events around every `every`-th GEMM launch from now on (1 = all)."""
        _lib.check(self.lib.vsr_profile_begin_sampled(self.h, int(every)))

    def profile_seen(self):
        return int(self.lib.vsr_profile_seen(self.h))

    def profile_bytes(self):
        """algorithmic bytes of the GEMM launches timed since profile_begin (read before profile_end)"""
        return float(self.lib.vsr_profile_bytes(self.h))

    @_on_device
    def profile_end(self, device):
        ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
        _lib.check(self.lib.vsr_profile_end(self.h, self._stream(device), C.byref(ms), C.byref(n), C.byref(fl)))
        return ms.value, n.value, fl.value

    # ------------------------------------------------------------------ loops
    @_on_device
    def greedy(self, B, device, verbs=None, gt=False):
        T = self.dims.seq_len
        words = torch.empty(B, T, dtype=torch.int64, device=device)
        gates = torch.empty(B, T, dtype=torch.int64, device=device)
        _lib.check(self.lib.vsr_greedy(self.h, _ptr(verbs), int(gt), _ptr(words), _ptr(gates), self._stream(device)))
        if verbs is not None:
            self.raise_on_bad_ids(device, "greedy (verbs)")
        return words, gates

    @_on_device
    def sample(self, B, device, seed, forced=None):
        T = self.dims.seq_len
        words = torch.empty(B, T, dtype=torch.int64, device=device)
        gates = torch.empty(B, T, dtype=torch.int64, device=device)
        lpw = torch.empty(B, T, dtype=torch.float32, device=device)
        lpg = torch.empty(B, T, dtype=torch.float32, device=device)
        fw = fg = None
        if forced is not None:
            fw = forced[0].to(device=device, dtype=torch.int64).contiguous()
            fg = forced[1].to(device=device, dtype=torch.int64).contiguous()
        _lib.check(self.lib.vsr_sample(self.h, C.c_uint64(seed), _ptr(fw), _ptr(fg), _ptr(words), _ptr(gates), _ptr(lpw),
                                       _ptr(lpg), self._stream(device)))
        if forced is not None:
            self.raise_on_bad_ids(device, "sample (forced ids)")
        return (words, gates), (lpw, lpg)

    @_on_device
    def beam(self, B, device, beam, out_size, eos_word, eos_gate, verbs=None, gt=False):
        T = self.dims.seq_len
        words = torch.empty(B, out_size, T, dtype=torch.int64, device=device)
        gates = torch.empty(B, out_size, T, dtype=torch.int64, device=device)
        lpw = torch.empty(B, out_size, T, dtype=torch.float32, device=device)
        lpg = torch.empty(B, out_size, T, dtype=torch.float32, device=device)
        scores = torch.empty(B, out_size, dtype=torch.float32, device=device)
        _lib.check(self.lib.vsr_beam(self.h, beam, out_size, int(eos_word), int(eos_gate), _ptr(verbs), int(gt), _ptr(words),
                                     _ptr(gates), _ptr(lpw), _ptr(lpg), _ptr(scores), self._stream(device)))
        if verbs is not None:
            self.raise_on_bad_ids(device, "beam (verbs)")
        return (words, gates), (lpw, lpg), scores

    @_on_device
    def xe_forward(self, B, device, captions):
        T = captions.size(1)
        V = self.dims.vocab_size
        captions = captions.to(device=device, dtype=torch.int64).contiguous()
        out = torch.empty(B, T, V, dtype=torch.float32, device=device)
        gate = torch.empty(B, T, 2, dtype=torch.float32, device=device)
        _lib.check(self.lib.vsr_xe_forward(self.h, _ptr(captions), T, _ptr(out), _ptr(gate), self._stream(device)))
        self.raise_on_bad_ids(device, "xe_forward")
        return out, gate

    @_on_device
    def step(self, t, rows_per_image, prev, state, verbs=None, gt=False):
        (h1, c1), (h2, c2), slot = state
        dev = h1.device
        M, V = h1.size(0), self.dims.vocab_size
        h1, c1, h2, c2 = (_f32(x, "state") for x in (h1, c1, h2, c2))
        slot = slot.to(torch.int64).contiguous()
        pw = pg = None
        if t > 0:
            pw = prev[0].to(torch.int64).contiguous()
            pg = prev[1].to(torch.int64).contiguous()
        outs = [torch.empty_like(h1) for _ in range(4)]
        slot_out = torch.empty_like(slot)
        lw = torch.empty(M, V, dtype=torch.float32, device=dev)
        lg = torch.empty(M, 2, dtype=torch.float32, device=dev)
        _lib.check(self.lib.vsr_step(self.h, t, rows_per_image, _ptr(pw), _ptr(pg), _ptr(h1), _ptr(c1), _ptr(h2), _ptr(c2),
                                     _ptr(slot), _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]), _ptr(outs[3]), _ptr(slot_out),
                                     _ptr(verbs), int(gt), _ptr(lw), _ptr(lg), self._stream(dev)))
        self.raise_on_bad_ids(dev, "step")
        return (lw, lg), ((outs[0], outs[1]), (outs[2], outs[3]), slot_out)
