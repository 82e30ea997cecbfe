"""Transformer pooling (poolings/transformer/transformer.py:6-9, transformer_module.py:27-117, poolings/common/transformer.py:9-33).

``Transformer_Module`` holds the parameters in torch containers of the reference's own structure (nn.Linear, ClsToken,
nn.TransformerEncoder) so that ``state_dict()`` keys, shapes and initialisation are the reference's and its checkpoints load
unchanged -- but their ``forward`` is never called: the arithmetic is ``ocrl_pool_transformer_fwd/_bwd`` (HIP), wrapped in a
``torch.autograd.Function`` because in the reference the pooling parameters belong to the RL policy's torch optimiser
(sb3s/ocr_extractor.py:33-35).  No CPU fallback: a CPU tensor raises."""
import ctypes
import math

import torch
from torch import nn

from .. import _lib
from .base import Base


class _ClsToken(nn.Module):
    def __init__(self, emb_size):
        super().__init__()
        self._cls_token = nn.Parameter(torch.zeros(emb_size))


class _PositionalEncoding(nn.Module):
    """poolings/common/transformer.py:60-82 (sin/cos * 0.001, a buffer) and :85-127 (stacked observations)"""

    def __init__(self, max_len, d_model, num_stacked_obss=1):
        super().__init__()
        if num_stacked_obss > 1:
            assert (max_len - 1) % num_stacked_obss == 0
            position = torch.arange((max_len - 1) // num_stacked_obss).repeat_interleave(num_stacked_obss)
            position = torch.cat([torch.tensor([0]), position + 1], dim=0).unsqueeze(1)
        else:
            position = torch.arange(max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(max_len, 1, d_model)
        pe[:, 0, 0::2] = torch.sin(position * div_term) * 0.001
        pe[:, 0, 1::2] = torch.cos(position * div_term) * 0.001
        self.register_buffer("pe", pe)


class _SinusoidalEncoding(nn.Module):
    """poolings/transformer/transformer_module.py:11-24: rows of the table se[i, 2j] = sin(i w_j), se[i, 2j+1] = cos(i w_j).  The reference
    materialises max_len + 1 rows as a buffer (`se`: 5 GB for push_embedding's max_len = 10^7); here the requested rows are evaluated
    on the fly (the same fp32 products), so the module has no `se` entry in its state_dict and ignores one when a reference checkpoint
    is loaded."""

    def __init__(self, d_model, max_len):
        super().__init__()
        self.d_model, self.max_len = d_model, max_len
        self.register_buffer("div_term", torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model)), persistent=False)

    def forward(self, idx):
        inp = idx.to(torch.float32).unsqueeze(-1)
        out = torch.empty(*idx.shape, self.d_model, device=idx.device, dtype=torch.float32)
        out[..., 0::2] = torch.sin(inp * self.div_term)
        out[..., 1::2] = torch.cos(inp * self.div_term)
        return out

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        state_dict.pop(prefix + "se", None)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class _LinearFn(torch.autograd.Function):
    """y = [relu](x W^T + b) on rows, forward and backward through the library's MFMA GEMM (include/ocrl_hip.h ocrl_gemm: NT with fused
    bias / ReLU, NN with the ReLU mask for dx, TN for dW) — the slot MLPs in front of the pooling transformer (transformer_module.py:47-63)"""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        if not x.is_cuda:
            raise RuntimeError("ocrl_amd.poolings: tensors must live on the GPU (there is no CPU fallback)")
        L = _lib.lib()
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).contiguous().float()
        w, b = weight.detach().contiguous(), bias.detach().contiguous()
        ctx.kpad = (-x2.shape[1]) % 4            # the GEMM wants a k extent that is a multiple of 4 (obj_emb of cw_embedding: 3*128 + 3 inputs)
        if ctx.kpad:
            x2 = torch.nn.functional.pad(x2, (0, ctx.kpad))
            w = torch.nn.functional.pad(w, (0, ctx.kpad))
        M, K = x2.shape
        N = weight.shape[0]
        y = torch.empty(M, N, device=x.device, dtype=torch.float32)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(L.ocrl_gemm(_lib.ptr(x2), _lib.ptr(w), _lib.ptr(y), M, N, K, K, K, N, 1, 1, 1.0, _lib.ptr(b), int(relu), None, 0, None, 0, 1, None, st))
        ctx.save_for_backward(x2, w, y)
        ctx.relu, ctx.shape = bool(relu), shp
        return y.reshape(*shp[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x2, w, y = ctx.saved_tensors
        M, K = x2.shape
        N = w.shape[0]
        dy2 = dy.reshape(M, N).contiguous().float()
        if ctx.relu:
            dy2 = dy2 * (y > 0)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, device=dy.device, dtype=torch.float32)           # dx = dy W        (NN)
            _lib.check(L.ocrl_gemm(_lib.ptr(dy2), _lib.ptr(w), _lib.ptr(dx), M, K, N, N, K, K, 1, 0, 1.0, None, 0, None, 0, None, 0, 1, None, st))
            if ctx.kpad:
                dx = dx[:, :K - ctx.kpad]
            dx = dx.reshape(ctx.shape)
        dw = torch.empty(N, K, device=dy.device, dtype=torch.float32)               # dW = dy^T x     (TN over the rows)
        _lib.check(L.ocrl_gemm(_lib.ptr(dy2), _lib.ptr(x2), _lib.ptr(dw), N, K, M, N, K, K, 0, 0, 1.0, None, 0, None, 0, None, 0, 1, None, st))
        if ctx.kpad:
            dw = dw[:, :K - ctx.kpad].contiguous()
        return dx, dw, dy2.sum(0), None


class _HipLinear(nn.Linear):
    """nn.Linear container (reference names / initialisation) whose arithmetic runs on the library's GEMM"""

    def __init__(self, i, o, relu=False):
        super().__init__(i, o)
        self._relu = relu

    def forward(self, x):
        return _LinearFn.apply(x, self.weight, self.bias, self._relu)


def _slot_mlp(widths):
    """nn.Sequential(Linear, ReLU, Linear[, ReLU, Linear]) with the reference's indices (the ReLUs are fused into the products, their
    slots are kept as Identity so that state_dict keys stay mlp.0 / mlp.2 / mlp.4)"""
    layers = []
    for k in range(len(widths) - 1):
        last = k == len(widths) - 2
        layers.append(_HipLinear(widths[k], widths[k + 1], relu=not last))
        if not last:
            layers.append(nn.Identity())
    return nn.Sequential(*layers)


class _PoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, slots, pos, geom, drop_p, seed, *params):
        if not slots.is_cuda:
            raise RuntimeError("ocrl_amd.poolings: tensors must live on the GPU (there is no CPU fallback)")
        L = _lib.lib()
        d, nhead, ff, nl = geom
        B, K, Din = slots.shape
        slots = slots.contiguous().float()
        ps = [p.detach().contiguous() for p in params]
        n = L.ocrl_pool_transformer_ws_floats(B, K, d, nhead, ff, nl)
        ws = torch.empty(n, device=slots.device, dtype=torch.float32)
        out = torch.empty(B, d, device=slots.device, dtype=torch.float32)
        arr = (ctypes.c_void_p * len(ps))(*[p.data_ptr() for p in ps])
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(L.ocrl_pool_transformer_fwd(_lib.ptr(slots), arr, _lib.ptr(pos), _lib.ptr(out), B, K, Din, d, nhead, ff, nl, drop_p, seed, _lib.ptr(ws), n, st))
        ctx.geom, ctx.drop_p, ctx.seed, ctx.ws, ctx.ps, ctx.slots = geom, drop_p, seed, ws, ps, slots
        ctx.need_dslots = ctx.needs_input_grad[0]      # read from the autograd node: the converted copy above carries no requires_grad
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        d, nhead, ff, nl = ctx.geom
        B, K, Din = ctx.slots.shape
        dout = dout.contiguous().float()
        gs = [torch.empty_like(p) for p in ctx.ps]
        ds = torch.empty_like(ctx.slots) if ctx.need_dslots else None
        arr = (ctypes.c_void_p * len(ctx.ps))(*[p.data_ptr() for p in ctx.ps])
        garr = (ctypes.c_void_p * len(gs))(*[g.data_ptr() for g in gs])
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(L.ocrl_pool_transformer_bwd(_lib.ptr(ctx.slots), _lib.ptr(dout), arr, _lib.ptr(ds), garr, B, K, Din, d, nhead, ff, nl, ctx.drop_p, ctx.seed,
                                               _lib.ptr(ctx.ws), ctx.ws.numel(), st))
        return (ds, None, None, None, None, *gs)


class _Transformer(nn.Module):
    """parameter container with the reference's names: _linear, _cls_token, _pos, _trans (poolings/common/transformer.py:9-19)"""

    def __init__(self, in_dim, d_model, nhead, num_layers, pos=None):
        super().__init__()
        self._linear = nn.Linear(in_dim, d_model)
        self._cls_token = _ClsToken(d_model)
        self._pos = pos
        self._trans = nn.TransformerEncoder(nn.TransformerEncoderLayer(d_model, nhead), num_layers, enable_nested_tensor=False)


class Transformer_Module(nn.Module):
    def __init__(self, ocr_rep_dim: int, ocr_num_slots: int, config, num_stacked_obss: int = 1) -> None:
        super().__init__()
        self.rep_dim = d_model = config.d_model
        self.config = config
        self.mlp = None
        if getattr(config, "use_mlp1", False):                  # transformer_module.py:47-53
            self.mlp = _slot_mlp([ocr_rep_dim, 64, 128])
            ocr_rep_dim = 128
        if getattr(config, "use_mlp2", False):                  # transformer_module.py:55-63 (replaces mlp1's module when both are set, as in the reference)
            self.mlp = _slot_mlp([ocr_rep_dim, 64, 64, 128])
            ocr_rep_dim = 128
        # embeddings of ground-truth state vectors (the ocr=GT encoders; transformer_module.py:65-78): quantised coordinates index a
        # sinusoid table, ids index nn.Embedding tables, the Linears to 128 run on the library's GEMM (gathers / concatenations are glue)
        self._cw = bool(getattr(config, "cw_embedding", False))
        self._push = bool(getattr(config, "push_embedding", False))
        if self._cw or self._push:
            if d_model != 128:
                raise ValueError("cw_embedding / push_embedding build 128-wide object embeddings: pooling.d_model must be 128 (transformer_module.py:65-78)")
        if self._cw:
            self.max_len = 10000
            self.cw_emb = _SinusoidalEncoding(d_model, self.max_len)
            self.arm_emb = _HipLinear(28 * d_model, 128)
            self.obj_emb = _HipLinear(3 * d_model + 3, 128)
            ocr_rep_dim = 128
        if self._push:
            self.max_len = 10000000
            self.color_emb = nn.Embedding(10, 128)
            self.shape_emb = nn.Embedding(10, 128)
            self.pos_emb = _SinusoidalEncoding(d_model, self.max_len)
            self.obj_emb = _HipLinear(4 * d_model, 128)
            ocr_rep_dim = 128
        if num_stacked_obss > 1:
            pos = _PositionalEncoding(ocr_num_slots * num_stacked_obss + 1, d_model, num_stacked_obss)
        elif config.pos_emb in ("ape", "lpe"):           # both map to the fixed table in the reference (transformer_module.py:40-43)
            pos = _PositionalEncoding(ocr_num_slots + 1, d_model)
        elif config.pos_emb == "None":
            pos = None
        else:
            raise ValueError(f"unknown pos_emb {config.pos_emb!r}")
        self._trans = _Transformer(ocr_rep_dim, d_model, config.nhead, config.num_layers, pos)
        layer = self._trans._trans.layers[0]
        self._geom = (d_model, config.nhead, layer.linear1.out_features, config.num_layers)
        self._drop_p = float(layer.dropout.p)
        self._calls = 0
        self.seed = 0

    def _param_list(self):
        t = self._trans
        ps = [t._linear.weight, t._linear.bias, t._cls_token._cls_token]
        for l in t._trans.layers:
            ps += [l.self_attn.in_proj_weight, l.self_attn.in_proj_bias, l.self_attn.out_proj.weight, l.self_attn.out_proj.bias,
                   l.linear1.weight, l.linear1.bias, l.linear2.weight, l.linear2.bias, l.norm1.weight, l.norm1.bias, l.norm2.weight, l.norm2.bias]
        return ps

    def get_pos_emb(self, emb, x):
        """transformer_module.py:82-87: [-1, 1] -> table row"""
        x = torch.clamp((x + 1) / 2, 0.0, 1.0)
        return emb((x // (1 / self.max_len)).long())

    def forward(self, state):
        if self._push:                                          # transformer_module.py:90-96
            color = self.color_emb(state[:, :, 0].long())
            shape = self.shape_emb(state[:, :, 1].long())
            pos = self.get_pos_emb(self.pos_emb, state[:, :, -2:])
            state = self.obj_emb(torch.cat([color, shape, pos[:, :, 0], pos[:, :, 1]], dim=-1))
        if self._cw:                                            # transformer_module.py:98-111
            B, K, _ = state.shape
            arm = self.arm_emb(self.get_pos_emb(self.cw_emb, state[:, 0, :28]).view(B, -1))
            obj = state[:, 1:, 28:]
            obj_pos = self.get_pos_emb(self.cw_emb, obj[:, :, :3].reshape(-1, 3)).reshape(B, K - 1, -1)
            objs = self.obj_emb(torch.cat([obj_pos, obj[:, :, 7:10]], dim=-1))
            state = torch.cat([arm.unsqueeze(1), objs], dim=1)
        if self.mlp is not None:                                # transformer_module.py:103-104
            state = self.mlp(state)
        pos = None if self._trans._pos is None else self._trans._pos.pe[: state.shape[1] + 1, 0].contiguous()
        p = self._drop_p if self.training else 0.0
        self._calls += 1
        seed = (int(self.seed) << 32) + self._calls           # a fresh dropout pattern per call, reproducible from `seed`
        return _PoolFn.apply(state, pos, self._geom, p, seed, *self._param_list())


class Transformer(Base):
    def __init__(self, ocr, config, num_stacked_obss: int = 1) -> None:
        self._module = Transformer_Module(ocr.rep_dim, ocr.num_slots, config, num_stacked_obss)
        super().__init__(ocr, config)
