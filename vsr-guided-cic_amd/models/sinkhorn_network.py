"""SinkhornNet (region ordering inside a semantic role) on MI355X: the reference's class surface over libvsrcap.so.

Mirrors /root/reference/models/sinkhorn_network.py:5-51: constructor (N, n_iters, tau), the five Linear layers (identical
state_dict keys, so eval_coco.py:102 loads 'saved_model/coco_sinkhorn/model-sh.pth'), forward(seq (b, N, 2352)) -> the
doubly-normalised (b, N, N) matrix.  `assign(seq)` additionally returns the assignment eval_coco.py:185-189 computes with
munkres on the host, for all items at once (vsr_sinkhorn_assign)."""
import torch
from torch import nn


class SinkhornNet(nn.Module):
    def __init__(self, N, n_iters, tau):
        super().__init__()
        self.N = N
        self.n_iters = n_iters
        self.tau = tau
        self.W1_txt = nn.Linear(300, 128)
        self.W1_vis = nn.Linear(2048, 512)
        self.W2_vis = nn.Linear(512, 128)
        self.W_fc_pos = nn.Linear(260, 256)
        self.W_fc = nn.Linear(256, N)
        self.init_weights()
        self._eng = None

    def init_weights(self):
        for m in (self.W1_txt, self.W1_vis, self.W2_vis, self.W_fc_pos, self.W_fc):
            nn.init.xavier_normal_(m.weight)
            nn.init.constant_(m.bias, 0)

    def _engine(self, device):
        if device.type != 'cuda':
            raise RuntimeError("SinkhornNet (MI355X build) computes only on the GPU: move the model and its inputs to 'cuda'. There is no CPU fallback.")
        from vsrcap.ssp import SspEngine
        key = tuple(p.data_ptr() for p in self.parameters())
        if self._eng is None or self._key != key:
            self._eng = SspEngine(device)
            self._eng.bind_sinkhorn({k: v.data for k, v in self.state_dict(keep_vars=True).items()}, self.N, self.n_iters, self.tau)
            self._key = key
        return self._eng

    def assign(self, seq):
        """seq (Q, N, 2352) -> (tr (Q,N,N), assign (Q,N) int64): assign[q][i] = column paired with row i of tr[q]^T"""
        eng = self._engine(seq.device)
        tr, a = eng.sinkhorn_assign(seq, want_matrix=True)
        return tr, a.long()

    def forward(self, seq):
        return self.assign(seq)[0]
