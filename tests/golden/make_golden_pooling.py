"""Generate the golden vectors that pin ``oracle/pooling_oracle.py`` to the reference's Transformer pooling.

Runs ONLY in the build container (needs /root/reference; import recipe = SURVEY.md Appendix C): imports the reference's own
``poolings.transformer.transformer_module.Transformer_Module``, loads closed-form weights, runs it on seeded slots and
(1) asserts the oracle restatement agrees, (2) writes small ``.npz`` fixtures (inputs, outputs, gradients) under tests/golden/.
Train-mode case: torch.nn.functional.dropout is replaced by a replay of given keep-masks (the three nn.Dropout sites of
nn.TransformerEncoderLayer, in call order dropout1, dropout, dropout2 per layer) and the attention-weight dropout, which lives inside
torch's fused attention and cannot be replayed, is switched off through the module's own ``self_attn.dropout`` attribute.

    python tests/golden/make_golden_pooling.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import pooling_oracle as PO  # noqa: E402


def import_reference():
    sys.path.insert(0, REF)
    for n in ("wandb", "h5py", "omegaconf"):
        sys.modules.setdefault(n, types.ModuleType(n))
    pkg = types.ModuleType("poolings")
    pkg.__path__ = [os.path.join(REF, "poolings")]
    sys.modules["poolings"] = pkg
    from poolings.transformer.transformer_module import Transformer_Module  # noqa
    return Transformer_Module


def ref_config(cfg):
    return types.SimpleNamespace(name="Transformer", rep_dim=cfg.d_model, d_model=cfg.d_model, nhead=cfg.nhead, num_layers=cfg.num_layers,
                                 pos_emb=cfg.pos_emb, norm_first=False, use_mlp1=False, use_mlp2=False, cw_embedding=False, push_embedding=False)


CASES = {
    "default": dict(over=dict(), B=3),                                                       # configs/pooling/transformer.yaml with SLATE's 6 x 192 slots
    "ape_l2": dict(over=dict(rep_dim=64, num_slots=4, num_layers=2, pos_emb="ape", nhead=4), B=2),
}


def run_case(Mod, tag, over, B, train):
    cfg = PO.default_cfg(**over)
    m = Mod(cfg.rep_dim, cfg.num_slots, ref_config(cfg))
    P = PO.formula_params(cfg)
    sd = m.state_dict()
    names = [n for n, _ in PO.param_shapes(cfg)]
    assert [k for k in sd if not k.endswith(".pe")] == names, "parameter inventory differs from the reference"
    m.load_state_dict({**sd, **P})
    g = torch.Generator().manual_seed(11)
    slots = torch.randn(B, cfg.num_slots, cfg.rep_dim, generator=g)
    cot = torch.randn(B, cfg.d_model, generator=g)
    S, d, ff = cfg.num_slots + 1, cfg.d_model, cfg.dim_feedforward
    masks, p_drop = None, 0.0
    orig = torch.nn.functional.dropout
    if train:
        p_drop = cfg.dropout
        masks, order = {}, []
        for l in range(cfg.num_layers):
            for key, shape in ((f"l{l}.drop1", (B, S, d)), (f"l{l}.ffn", (B, S, ff)), (f"l{l}.drop2", (B, S, d))):
                masks[key] = (torch.rand(shape, generator=g) >= p_drop).float()
                order.append(key)
            m._trans._trans.layers[l].self_attn.dropout = 0.0
        it = iter(order)

        def replay(x, p=0.5, training=True, inplace=False):
            k = next(it)
            mk = masks[k].permute(1, 0, 2)                  # the reference runs [S,B,*]
            assert mk.shape == x.shape and abs(p - p_drop) < 1e-9 and training, (k, mk.shape, x.shape)
            return x * mk / (1.0 - p)
        torch.nn.functional.dropout = replay
        m.train()
    else:
        m.eval()
    try:
        s = slots.clone().requires_grad_(True)
        # keep torch on its python path (the fused inference fast-path is a different kernel with the same maths)
        out = m(s)
        (out * cot).sum().backward()
    finally:
        torch.nn.functional.dropout = orig
    ref_g = {n: p.grad.clone() for n, p in m.named_parameters()}
    o_out, o_g, o_ds = PO.loss_and_grads(P, slots, cfg, cot, masks, p_drop)
    err = (o_out - out.detach()).abs().max().item() / out.detach().abs().max().item()
    assert err < 2e-5, (tag, "out", err)
    for n in names:
        e = (o_g[n] - ref_g[n]).abs().max().item() / max(ref_g[n].abs().max().item(), 1e-6 * max(v.abs().max().item() for v in ref_g.values()))
        assert e < 2e-4, (tag, n, e)
    e = (o_ds - s.grad).abs().max().item() / s.grad.abs().max().item()
    assert e < 2e-4, (tag, "dslots", e)
    print(f"[{tag}{'/train' if train else ''}] oracle == reference (out {err:.1e})")
    fx = {"slots": slots.numpy(), "cot": cot.numpy(), "out": out.detach().numpy(), "dslots": s.grad.numpy(),
          "cfg": np.array([cfg.rep_dim, cfg.num_slots, cfg.d_model, cfg.nhead, cfg.num_layers, cfg.dim_feedforward, int(cfg.pos_emb != "None"), B]),
          "p_drop": np.array([p_drop], dtype=np.float32)}
    for n in names:                      # gradients: three moments + a strided sample keep the fixture small
        t = ref_g[n].double().flatten()
        fx["g:" + n] = np.concatenate([np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()]), t[:: max(1, t.numel() // 509)][:509].numpy()])
    if train:
        for k, v in masks.items():
            fx["m:" + k] = np.packbits(v.numpy().astype(np.uint8).reshape(-1))
    return fx


def gt_states(mode, B, K, seed):
    """seeded ground-truth state vectors in the ranges the embeddings index (transformer_module.py:82-111)"""
    g = torch.Generator().manual_seed(seed)
    if mode == "push":          # [colour id, shape id, ..., x, y]
        return torch.cat([torch.randint(0, 10, (B, K, 2), generator=g).float(), torch.rand(B, K, 2, generator=g) * 2.4 - 1.2], dim=-1)
    return torch.rand(B, K, 28 + 10, generator=g) * 2.4 - 1.2      # row 0: 28 arm coordinates; rows 1..: [28:31] position, [35:38] colour


def run_gt_case(Mod, mode):
    """cw_embedding / push_embedding (transformer_module.py:65-111): the reference module on seeded states, closed-form weights"""
    cfg = PO.default_cfg(rep_dim=128, num_slots=5, d_model=128)
    rc = ref_config(cfg)
    setattr(rc, "cw_embedding" if mode == "cw" else "push_embedding", True)
    m = Mod(cfg.rep_dim, cfg.num_slots, rc)          # push: builds the reference's 10^7-row sinusoid table (5 GB, tens of seconds)
    P, G = PO.formula_params(cfg), PO.gt_formula_params(mode)
    sd = m.state_dict()
    extra = [k for k in sd if not k.startswith("_trans.") and not k.endswith(".se")]
    assert sorted(extra) == sorted(G), (extra, sorted(G))
    m.load_state_dict({**sd, **P, **G})
    m.eval()
    B, K = 3, cfg.num_slots
    state = gt_states(mode, B, K, 21)
    cot = torch.randn(B, cfg.d_model, generator=torch.Generator().manual_seed(22))
    out = m(state)
    (out * cot).sum().backward()
    ref_g = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    # oracle: embedding restatement + the pinned transformer restatement
    Gq = {k: v.clone().requires_grad_(True) for k, v in G.items()}
    Pq = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    emb = PO.gt_embed(Gq, state, mode)
    o = PO.forward(Pq, emb, cfg)
    (o * cot).sum().backward()
    err = (o.detach() - out.detach()).abs().max().item() / out.detach().abs().max().item()
    assert err < 2e-5, (mode, err)
    gmax = max(v.abs().max().item() for v in ref_g.values())
    for n in G:
        e = (Gq[n].grad - ref_g[n]).abs().max().item() / max(ref_g[n].abs().max().item(), 1e-6 * gmax)
        assert e < 2e-4, (mode, n, e)
    print(f"[{mode}_embedding] oracle == reference (out {err:.1e})")
    fx = {"state": state.numpy(), "cot": cot.numpy(), "out": out.detach().numpy(), "emb": emb.detach().numpy()}
    for n in G:
        t = ref_g[n].double().flatten()
        fx["g:" + n] = np.concatenate([np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()]), t[:: max(1, t.numel() // 509)][:509].numpy()])
    np.savez_compressed(os.path.join(HERE, f"pooling_{mode}.npz"), **fx)


def main():
    Mod = import_reference()
    torch.manual_seed(0)
    if "--only-gt" in sys.argv:
        run_gt_case(Mod, "cw")
        run_gt_case(Mod, "push")
        return
    for tag, c in CASES.items():
        np.savez_compressed(os.path.join(HERE, f"pooling_{tag}.npz"), **run_case(Mod, tag, c["over"], c["B"], False))
    np.savez_compressed(os.path.join(HERE, "pooling_default_train.npz"), **run_case(Mod, "default", CASES["default"]["over"], 2, True))
    run_gt_case(Mod, "cw")
    run_gt_case(Mod, "push")


if __name__ == "__main__":
    main()
