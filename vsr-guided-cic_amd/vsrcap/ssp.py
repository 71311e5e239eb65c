"""Host driver of the ordering-model entry points of libvsrcap.so (include/vsrcap.h: vsr_ssp_*, vsr_sinkhorn_assign): one
vsr_ssp object per device, weights borrowed from torch parameters, work enqueued on torch's current stream.  No fallback:
CPU tensors raise."""
import ctypes as C

import torch

from . import _lib


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _need_gpu(t, name):
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU (got %s); this path has no CPU implementation" % (name, t.device))


class SspEngine:
    def __init__(self, device):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.vsr_ssp_create(C.byref(self.h)))
        self._ws = None
        self._keep = {}

    def __del__(self):
        try:
            if self.h:
                self.lib.vsr_ssp_destroy(self.h)
                self.h = C.c_void_p()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _workspace(self, need):
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    @staticmethod
    def _f32(sd, key):
        t = sd[key]
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
            raise RuntimeError("weight %s must be a contiguous fp32 GPU tensor (got %s %s)" % (key, t.dtype, t.device))
        return t.data_ptr()

    def bind_ssp(self, sd):
        """sd: state_dict-like mapping with the reference S_SSP's keys (models/sort_model.py)."""
        def layer(pre, dec):
            vals = {}
            for i in (1, 2, 3):
                for wb, f in (("weight", "w"), ("bias", "b")):
                    vals["ln%d_%s" % (i, f)] = self._f32(sd, "%s.layer_norm%d.%s" % (pre, i, wb)) if (i < 3 or dec) else 0
            for q in "QKVO":
                vals["W" + q.lower()] = self._f32(sd, "%s.attention.linear_%s.weight" % (pre, q))
                vals["b" + q.lower()] = self._f32(sd, "%s.attention.linear_%s.bias" % (pre, q))
            vals["W1"], vals["b1"] = self._f32(sd, pre + ".ff_layer.w_1.weight"), self._f32(sd, pre + ".ff_layer.w_1.bias")
            vals["W2"], vals["b2"] = self._f32(sd, pre + ".ff_layer.w_2.weight"), self._f32(sd, pre + ".ff_layer.w_2.bias")
            return _lib.VsrSspLayer(**vals)
        w = _lib.VsrSspWeights()
        w.sr_embed, w.v_embed = self._f32(sd, "sr_embed_layer.weight"), self._f32(sd, "v_embed_layer.weight")
        w.n_verbs = sd["v_embed_layer.weight"].shape[0]
        w.fc_w, w.fc_b = self._f32(sd, "encoder.fc_feat.weight"), self._f32(sd, "encoder.fc_feat.bias")
        for l in range(3):
            w.enc[l] = layer("encoder.encoder_layers.%d" % l, False)
            w.dec[l] = layer("decoder.encoder_layers.%d" % l, True)
        w.enc_ln_w, w.enc_ln_b = self._f32(sd, "encoder.layer_norm.weight"), self._f32(sd, "encoder.layer_norm.bias")
        w.dec_ln_w, w.dec_ln_b = self._f32(sd, "decoder.layer_norm.weight"), self._f32(sd, "decoder.layer_norm.bias")
        w.exp_w, w.exp_b = self._f32(sd, "expander_nn.weight"), self._f32(sd, "expander_nn.bias")
        with torch.cuda.device(self.device):
            _lib.check(self.lib.vsr_ssp_bind(self.h, C.byref(w), None))
        self._keep["ssp"] = sd

    def bind_sinkhorn(self, sd, N, n_iters, tau):
        w = _lib.VsrSinkhornWeights()
        for f in _lib.SINKHORN_FIELDS:
            name, wb = f.rsplit("_", 1)
            setattr(w, f, self._f32(sd, "%s.%s" % (name, "weight" if wb == "w" else "bias")))
        w.N, w.n_iters, w.tau = int(N), int(n_iters), float(tau)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.vsr_ssp_bind(self.h, None, C.byref(w)))
        self._keep["sinkhorn"] = sd
        self.N = int(N)

    def generate(self, verbs, roles):
        """verbs (S,) int64, roles (S,10) int (0 = padding) on the GPU -> pred (S,10) int32, logp (S,10) fp32"""
        _need_gpu(verbs, "verbs")
        _need_gpu(roles, "roles")
        verbs = verbs.to(torch.int64).contiguous()
        roles = roles.to(torch.int32).contiguous()
        S = roles.size(0)
        if roles.dim() != 2 or roles.size(1) != 10 or verbs.numel() != S:
            raise RuntimeError("expected verbs (S,) and roles (S,10); got %s and %s" % (tuple(verbs.shape), tuple(roles.shape)))
        # the reference's embeddings raise IndexError for ids outside their tables (sort_model.py:108); the kernels would clamp
        lo, hi = int(roles.min()), int(roles.max())
        if lo < 0 or hi >= 26:
            raise IndexError("semantic-role ids must lie in [0, 26) (0 = padding); got [%d, %d]" % (lo, hi))
        pred = torch.empty(S, 10, dtype=torch.int32, device=self.device)
        logp = torch.empty(S, 10, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            ws = self._workspace(self.lib.vsr_ssp_workspace_bytes(S))
            _lib.check(self.lib.vsr_ssp_generate(self.h, _ptr(verbs), _ptr(roles), S, _ptr(pred), _ptr(logp), _ptr(ws), ws.numel(), self._stream()))
        return pred, logp

    def sinkhorn_assign(self, seq, want_matrix=True):
        """seq (Q,N,2352) fp32 on the GPU -> (tr (Q,N,N) or None, assign (Q,N) int32)"""
        _need_gpu(seq, "seq")
        seq = seq.float().contiguous()
        Q = seq.size(0)
        if seq.dim() != 3 or seq.size(1) != self.N or seq.size(2) != 2352:
            raise RuntimeError("expected (Q, %d, 2352) rows; got %s" % (self.N, tuple(seq.shape)))
        tr = torch.empty(Q, self.N, self.N, dtype=torch.float32, device=self.device) if want_matrix else None
        assign = torch.empty(Q, self.N, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            ws = self._workspace(self.lib.vsr_sinkhorn_workspace_bytes(Q, self.N))
            _lib.check(self.lib.vsr_sinkhorn_assign(self.h, _ptr(seq), Q, _ptr(tr), _ptr(assign), _ptr(ws), ws.numel(), self._stream()))
        return tr, assign
