"""TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg): CPU restatement of the reference's
slot-set pooling head, the consumer of the slots on the RL side.

Follows /root/reference:
  poolings/common/transformer.py:9-33     Transformer: Linear(rep_dim -> d_model); prepend the CLS parameter; permute to [S,B,d]; optional
                                          positional table; nn.TransformerEncoder(nn.TransformerEncoderLayer(d_model, nhead), num_layers);
                                          row 0 (CLS) of the output.  `norm_first` is accepted and ignored by the reference (:15-17).
  poolings/common/transformer.py:60-82    PositionalEncoding: sin/cos table scaled by 0.001 (used for BOTH "ape" and "lpe",
                                          poolings/transformer/transformer_module.py:40-43)
  poolings/transformer/transformer_module.py:27-117   Transformer_Module with its default switches (no mlp / cw / push embeddings)
The encoder layer itself is a third-party dependency (torch.nn.TransformerEncoderLayer; the reference pins no torch version, this image
has torch 2.10): post-norm, ReLU, dim_feedforward 2048, dropout 0.1, layer_norm_eps 1e-5, batch_first False:
  x = LN1(x + drop1(MHA(x)));  x = LN2(x + drop2(W2 drop(relu(W1 x + b1)) + b2))
  MHA: qkv = x Win^T + bin; heads of d/nhead; softmax(q k^T / sqrt(hd)); dropout on the weights; concat; out_proj.
It is restated here with explicit tensor algebra (no nn.TransformerEncoderLayer call) so that autograd provides the gradient oracle and
dropout keep-masks can be injected.  Pinned by tests/golden/pooling_*.npz (made by tests/golden/make_golden_pooling.py from the
reference module itself): eval mode outputs and all gradients, and train mode with the three nn.Dropout sites replayed.  The
attention-weight dropout sits inside torch's fused scaled_dot_product_attention and cannot be replayed: its placement (on the softmax
output, scaled by 1/(1-p)) follows torch's documentation -- parity unpinned for that one site.
"""
import math
import types

import torch


def default_cfg(**over):
    c = types.SimpleNamespace(rep_dim=192, num_slots=6, d_model=128, nhead=8, num_layers=1, dim_feedforward=2048, dropout=0.1, pos_emb="None")
    for k, v in over.items():
        assert hasattr(c, k), k
        setattr(c, k, v)
    return c


def param_shapes(cfg):
    """state_dict order of Transformer_Module (names relative to the module)"""
    d, ff = cfg.d_model, cfg.dim_feedforward
    out = [("_trans._linear.weight", (d, cfg.rep_dim)), ("_trans._linear.bias", (d,)), ("_trans._cls_token._cls_token", (d,))]
    for l in range(cfg.num_layers):
        p = f"_trans._trans.layers.{l}."
        out += [(p + "self_attn.in_proj_weight", (3 * d, d)), (p + "self_attn.in_proj_bias", (3 * d,)),
                (p + "self_attn.out_proj.weight", (d, d)), (p + "self_attn.out_proj.bias", (d,)),
                (p + "linear1.weight", (ff, d)), (p + "linear1.bias", (ff,)), (p + "linear2.weight", (d, ff)), (p + "linear2.bias", (d,)),
                (p + "norm1.weight", (d,)), (p + "norm1.bias", (d,)), (p + "norm2.weight", (d,)), (p + "norm2.bias", (d,))]
    return out


def formula_params(cfg, gain=1.0):
    """closed-form fp32 weights (no RNG): reproducible in the golden generator and on the GPU box"""
    P = {}
    for i, (name, shape) in enumerate(param_shapes(cfg)):
        n = 1
        for s in shape:
            n *= s
        k = torch.arange(n, dtype=torch.float64)
        v = torch.sin(k * (0.37 + 0.011 * i) + 0.5 * i)
        if len(shape) == 2:
            v = v * gain / math.sqrt(shape[1])
        elif name.endswith("norm1.weight") or name.endswith("norm2.weight"):
            v = 1.0 + 0.1 * v
        else:
            v = 0.1 * v
        P[name] = v.reshape(shape).float()
    return P


def pos_table(cfg):
    """[S, d] table added to the token sequence, or None (transformer_module.py:38-45, transformer.py:60-82)"""
    if cfg.pos_emb == "None":
        return None
    S, d = cfg.num_slots + 1, cfg.d_model
    position = torch.arange(S).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d, 2) * (-math.log(10000.0) / d))
    pe = torch.zeros(S, d)
    pe[:, 0::2] = torch.sin(position * div_term) * 0.001
    pe[:, 1::2] = torch.cos(position * div_term) * 0.001
    return pe


def _ln(x, g, b):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + 1e-5) * g + b


def forward(P, slots, cfg, masks=None, p_drop=0.0):
    """slots [B,K,rep_dim] -> [B,d_model].  masks: {"l{i}.attn" [B,h,S,S], "l{i}.drop1" [B,S,d], "l{i}.ffn" [B,S,ff], "l{i}.drop2" [B,S,d]}
    keep-masks (1 = kept) applied with scale 1/(1-p_drop); None = eval mode."""
    B, K, _ = slots.shape
    d, h = cfg.d_model, cfg.nhead
    hd, S = d // h, K + 1

    def drop(x, key):
        if masks is None or p_drop == 0.0 or key not in masks:      # a site without a mask is not dropped
            return x
        return x * masks[key].to(x.dtype) / (1.0 - p_drop)

    x = slots @ P["_trans._linear.weight"].T + P["_trans._linear.bias"]
    x = torch.cat([P["_trans._cls_token._cls_token"].reshape(1, 1, d).expand(B, 1, d), x], 1)        # [B,S,d] (batch-major; per-sample maths)
    pe = pos_table(cfg)
    if pe is not None:
        x = x + pe.to(x.dtype)
    for l in range(cfg.num_layers):
        p = f"_trans._trans.layers.{l}."
        qkv = x @ P[p + "self_attn.in_proj_weight"].T + P[p + "self_attn.in_proj_bias"]
        q, k, v = [t.reshape(B, S, h, hd).permute(0, 2, 1, 3) for t in qkv.split(d, -1)]
        w = torch.softmax((q * hd ** -0.5) @ k.transpose(-1, -2), -1)
        o = (drop(w, f"l{l}.attn") @ v).permute(0, 2, 1, 3).reshape(B, S, d)
        a = o @ P[p + "self_attn.out_proj.weight"].T + P[p + "self_attn.out_proj.bias"]
        x = _ln(x + drop(a, f"l{l}.drop1"), P[p + "norm1.weight"], P[p + "norm1.bias"])
        f = drop(torch.relu(x @ P[p + "linear1.weight"].T + P[p + "linear1.bias"]), f"l{l}.ffn")
        f = f @ P[p + "linear2.weight"].T + P[p + "linear2.bias"]
        x = _ln(x + drop(f, f"l{l}.drop2"), P[p + "norm2.weight"], P[p + "norm2.bias"])
    return x[:, 0]


# ---- embeddings of ground-truth state vectors in front of the transformer (poolings/transformer/transformer_module.py:65-101, the
# cw_embedding / push_embedding switches of ocr=GT): sinusoidal tables indexed by quantised coordinates, two nn.Embedding tables, one or
# two Linears to 128.  Pinned by tests/golden/pooling_cw.npz / pooling_push.npz (made from the reference module).
def sinusoid(idx, d_model):
    """rows `idx` of the reference's SinusoidalEncoding table (transformer_module.py:11-24): se[i, 2j] = sin(i w_j), se[i, 2j+1] = cos(i w_j)"""
    inp = idx.to(torch.float32).unsqueeze(-1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    out = torch.zeros(*idx.shape, d_model, dtype=torch.float32)
    out[..., 0::2] = torch.sin(inp * div_term)
    out[..., 1::2] = torch.cos(inp * div_term)
    return out


def quantise(x, max_len):
    """get_pos_emb (transformer_module.py:82-87): [-1, 1] -> table index"""
    x = torch.clamp((x + 1) / 2, 0.0, 1.0)
    return (x // (1 / max_len)).long()


def gt_param_shapes(mode, d_model=128):
    if mode == "cw":
        return [("arm_emb.weight", (128, 28 * d_model)), ("arm_emb.bias", (128,)), ("obj_emb.weight", (128, 3 * d_model + 3)), ("obj_emb.bias", (128,))]
    return [("color_emb.weight", (10, 128)), ("shape_emb.weight", (10, 128)), ("obj_emb.weight", (128, 4 * d_model)), ("obj_emb.bias", (128,))]


def gt_formula_params(mode, d_model=128):
    P = {}
    for i, (name, shape) in enumerate(gt_param_shapes(mode, d_model)):
        n = 1
        for s in shape:
            n *= s
        v = torch.sin(torch.arange(n, dtype=torch.float64) * (0.23 + 0.017 * i) + 0.9 * i)
        if len(shape) == 2 and (name.startswith("arm_emb") or name.startswith("obj_emb")):
            v = v / math.sqrt(shape[1])          # Linear weights
        else:
            v = 0.5 * v                          # embedding tables, biases
        P[name] = v.reshape(shape).float()
    return P


def gt_embed(G, state, mode, d_model=128):
    """state [B,K,*] of ground-truth vectors -> [B,K,128] (the tensor the transformer's input Linear consumes)"""
    if mode == "push":                      # transformer_module.py:90-96
        color = G["color_emb.weight"][state[:, :, 0].long()]
        shape = G["shape_emb.weight"][state[:, :, 1].long()]
        pos = sinusoid(quantise(state[:, :, -2:], 10000000), d_model).to(state.dtype)
        x = torch.cat([color, shape, pos[:, :, 0], pos[:, :, 1]], dim=-1)
        return x @ G["obj_emb.weight"].T + G["obj_emb.bias"]
    B, K, _ = state.shape                   # transformer_module.py:98-111
    arm = sinusoid(quantise(state[:, 0, :28], 10000), d_model).to(state.dtype).reshape(B, -1)
    arm = arm @ G["arm_emb.weight"].T + G["arm_emb.bias"]
    obj = state[:, 1:, 28:]
    opos = sinusoid(quantise(obj[:, :, :3].reshape(-1, 3), 10000), d_model).to(state.dtype).reshape(B, K - 1, -1)
    ob = torch.cat([opos, obj[:, :, 7:10]], dim=-1) @ G["obj_emb.weight"].T + G["obj_emb.bias"]
    return torch.cat([arm.unsqueeze(1), ob], dim=1)


def loss_and_grads(P, slots, cfg, cot, masks=None, p_drop=0.0, dtype=torch.float32):
    """out, d<out, cot>/dP, d<out, cot>/dslots"""
    Q = {k: v.detach().to(dtype).requires_grad_(True) for k, v in P.items()}
    s = slots.detach().to(dtype).requires_grad_(True)
    out = forward(Q, s, cfg, masks, p_drop)
    (out * cot.to(dtype)).sum().backward()
    return out.detach(), {k: v.grad for k, v in Q.items()}, s.grad
