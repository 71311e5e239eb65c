"""S_SSP (semantic-role ordering transformer) on MI355X: the reference's class surface over libvsrcap.so.

Mirrors /root/reference/models/sort_model.py:13-52 (constructor, module tree -> identical state_dict keys and shapes, so
`re_sort_net.load_state_dict(torch.load('saved_model/coco_s_ssp/model-tr.pth'))`, eval_coco.py:96, works) and
:105-183 `generate(this_verb, det_seqs_sr, mode='not-normal')`, the call of eval_coco.py:174.  All compute is the batched
HIP path (vsr_ssp_generate); `generate_batch` takes every (caption, verb) sequence of a loader batch in ONE call.
Training the ordering model (forward / loss, coco_scripts/train_region_sort.py) and the free-running 'normal' decode are
outside the hot path and raise."""
import math

import torch
from torch import nn


class _PositionalEmbedding(nn.Module):                      # transformer_modules.py:272-300 (buffer only; pos_enc=False in eval)
    def __init__(self, size, max_len=5000):
        super().__init__()
        pe = torch.zeros(max_len, size)
        position = torch.arange(0, max_len).unsqueeze(1).float()
        div_term = torch.exp((torch.arange(0, size, 2).float() * -(math.log(10000.0) / size)).float())
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer('pe', pe.unsqueeze(0))


class _TransformerEmbedding(nn.Embedding):                  # transformer_modules.py:182-215
    def __init__(self, num_embeddings, embedding_dim):
        super().__init__(num_embeddings, embedding_dim)
        self.pos_layer = _PositionalEmbedding(embedding_dim)


class _MultiHeadAttention(nn.Module):                       # transformer_modules.py:67-96
    def __init__(self, size):
        super().__init__()
        self.linear_Q = nn.Linear(size, size)
        self.linear_K = nn.Linear(size, size)
        self.linear_V = nn.Linear(size, size)
        self.linear_O = nn.Linear(size, size)


class _FeedForward(nn.Module):                              # transformer_modules.py:302-319
    def __init__(self, size, hidden):
        super().__init__()
        self.w_1 = nn.Linear(size, hidden)
        self.w_2 = nn.Linear(hidden, size)


class _EncoderLayer(nn.Module):                             # transformer_modules.py:321-345
    def __init__(self, size):
        super().__init__()
        self.attention = _MultiHeadAttention(size)
        self.ff_layer = _FeedForward(size, 4 * size)
        self.layer_norm1 = nn.LayerNorm(size)
        self.layer_norm2 = nn.LayerNorm(size)


class _DecoderLayer(nn.Module):                             # sort_modules.py:65-98 (cross_attention exists but is never called)
    def __init__(self, size):
        super().__init__()
        self.attention = _MultiHeadAttention(size)
        self.cross_attention = _MultiHeadAttention(size)
        self.ff_layer = _FeedForward(size, 4 * size)
        self.layer_norm1 = nn.LayerNorm(size)
        self.layer_norm2 = nn.LayerNorm(size)
        self.layer_norm3 = nn.LayerNorm(size)


class _Encoder(nn.Module):                                  # sort_modules.py:25-62
    def __init__(self, sr_embed_layer, v_embed_layer, size, n_layers):
        super().__init__()
        self.sr_embed_layer = sr_embed_layer
        self.v_embed_layer = v_embed_layer
        self.layer_norm = nn.LayerNorm(size)
        self.encoder_layers = nn.ModuleList([_EncoderLayer(size) for _ in range(n_layers)])
        self.fc_feat = nn.Linear(512, 512)


class _Decoder(nn.Module):                                  # sort_modules.py:101-135
    def __init__(self, embed_layer, size, n_layers):
        super().__init__()
        self.embed_layer = embed_layer
        self.layer_norm = nn.LayerNorm(size)
        self.encoder_layers = nn.ModuleList([_DecoderLayer(size) for _ in range(n_layers)])


class _LabelSmooth(nn.Module):                              # transformer_modules.py:150-165 (buffer only: part of the state_dict)
    def __init__(self, label_smoothing, n):
        super().__init__()
        self.register_buffer('one_hot', torch.full((n,), label_smoothing / (n - 2)).unsqueeze(0))


class S_SSP(nn.Module):
    def __init__(self, pos_enc=False, add_fc=True, dataset='coco'):
        super().__init__()
        if pos_enc or not add_fc:
            raise NotImplementedError("the MI355X build implements the configuration the eval scripts use: S_SSP() = pos_enc=False, add_fc=True")
        torch.manual_seed(1234)
        self._verb_size = 2662 if dataset == 'coco' else 2926
        self.encoder_layers = 3
        self.decoder_layers = 3
        self.max_len = 10
        self.beam_size = 1
        self.hidden_size = 512
        self.embed_size = 512
        self.sr_embed_layer = _TransformerEmbedding(26, self.embed_size)
        self.v_embed_layer = _TransformerEmbedding(self._verb_size + 1, self.embed_size)
        self.encoder = _Encoder(self.sr_embed_layer, self.v_embed_layer, self.hidden_size, self.encoder_layers)
        self.decoder = _Decoder(self.sr_embed_layer, self.hidden_size, self.decoder_layers)
        self.expander_nn = nn.Linear(self.hidden_size, 26)
        self.label_smooth = _LabelSmooth(0.1, 26)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        self._eng = None
        self.last_logprobs = None

    def _engine(self, device):
        if device.type != 'cuda':
            raise RuntimeError("S_SSP (MI355X build) computes only on the GPU: move the model and its inputs to 'cuda'. There is no CPU fallback.")
        from vsrcap.ssp import SspEngine
        key = tuple(p.data_ptr() for p in self.parameters())
        if self._eng is None or self._eng.device.index != (device.index if device.index is not None else torch.cuda.current_device()) or self._key != key:
            self._eng = SspEngine(device)
            self._eng.bind_ssp({k: v.data for k, v in self.state_dict(keep_vars=True).items()})
            self._key = key
        return self._eng

    def generate_batch(self, verbs, roles):
        """S sequences at once: verbs (S,), roles (S,10) with 0 = padding -> (pred (S,10) int64, log-probs (S,10) fp32)."""
        dev = self.expander_nn.weight.device
        eng = self._engine(dev)
        pred, logp = eng.generate(torch.as_tensor(verbs).to(dev).reshape(-1), torch.as_tensor(roles).to(dev))
        return pred.long(), logp

    def generate(self, this_verb, det_seqs_sr, mode='normal'):
        """eval_coco.py:174: output = re_sort_net.generate(this_verb (1,), verb_det_seqs_sr (1,10), mode='not-normal')."""
        if mode == 'normal':
            raise NotImplementedError("free-running 'normal' decoding is not on the eval path (eval_coco.py:174 uses mode='not-normal')")
        pred, logp = self.generate_batch(this_verb, det_seqs_sr)
        self.last_logprobs = logp
        # the reference allocates its log-prob buffer with det_seqs_sr.new_zeros (:121): an integer tensor, values truncated
        return pred, logp.trunc().to(pred.dtype), None

    def forward(self, *a, **k):
        raise NotImplementedError("training S_SSP (coco_scripts/train_region_sort.py) is outside the hot path (SURVEY.md section 2 row 11)")
