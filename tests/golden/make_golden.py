"""Generate the golden vectors that pin ``oracle/slate_oracle.py`` to the reference.

Runs ONLY in the build container (needs /root/reference; recipe = SURVEY.md Appendix C).
It imports the reference's own ``ocrs.slate.slate.SLATE``, loads closed-form weights,
injects the same noise, runs ``get_loss``/``backward``/``update`` and
  (1) asserts the oracle restatement agrees with the reference, and
  (2) writes small ``.npz`` fixtures (inputs are re-derivable from seeds; expected outputs
      are stored) under tests/golden/ for tests/test_oracle_golden.py.
The fixtures are data only: no reference source text is stored.

    python tests/golden/make_golden.py
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


def import_reference():
    sys.path.insert(0, REF)
    for n in ("wandb", "h5py", "omegaconf"):
        sys.modules.setdefault(n, types.ModuleType(n))
    pkg = types.ModuleType("ocrs")
    pkg.__path__ = [os.path.join(REF, "ocrs")]
    sys.modules["ocrs"] = pkg
    from ocrs.slate.slate import SLATE  # noqa
    return SLATE


def ref_config(cfg):
    NS = types.SimpleNamespace
    ocr = NS(
        name="SLATE", tau_start=cfg.tau_start, tau_final=cfg.tau_final, tau_steps=cfg.tau_steps,
        hard=cfg.hard, use_cnn_feat=False, use_bcdec=cfg.use_bcdec,
        dvae=NS(vocab_size=cfg.vocab_size, d_model=cfg.d_model),
        cnn=NS(hidden_size=cfg.cnn_hidden),
        slotattr=NS(num_iterations=cfg.num_iterations, num_slots=cfg.num_slots,
                    num_slot_heads=cfg.num_slot_heads, slot_size=cfg.slot_size,
                    mlp_hidden_size=cfg.mlp_hidden, pos_channels=4),
        tfdec=NS(num_dec_blocks=cfg.num_dec_blocks, num_dec_heads=cfg.num_dec_heads),
        learning=NS(lr_half_life=cfg.lr_half_life, lr_dvae=cfg.lr_dvae, lr_enc=cfg.lr_enc,
                    lr_dec=cfg.lr_dec, lr_warmup_steps=cfg.lr_warmup_steps,
                    dropout=cfg.dropout, clip=cfg.clip),
    )
    env = NS(obs_size=cfg.obs_size, obs_channels=cfg.obs_channels)
    return ocr, env


class DropoutReplay:
    """Monkey-patch torch.nn.functional.dropout to consume recorded keep-masks in call order."""

    def __init__(self, masks_in_order, p):
        self.masks = list(masks_in_order)
        self.p = p
        self.i = 0

    def __call__(self, x, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return x
        m = self.masks[self.i]
        self.i += 1
        assert m.shape == x.shape, (m.shape, x.shape)
        return x * m / (1.0 - p)


def mask_order(cfg):
    keys = ["z_pos"]
    for b in range(cfg.num_dec_blocks):
        keys += [f"blk{b}.self.attn", f"blk{b}.self.out", f"blk{b}.cross.attn", f"blk{b}.cross.out", f"blk{b}.ffn"]
    return keys


def summarize(t):
    t = t.detach().double().flatten()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


def run_case(tag, cfg, B, seed, train_dropout, n_steps, SLATE, full, grad_tol=1e-3):
    import torch.nn.functional as F
    from oracle import slate_oracle as O

    torch.manual_seed(0)
    ocr, env = ref_config(cfg)
    model = SLATE(ocr, env)
    P = O.formula_params(cfg)
    sd = model._module.state_dict()
    for k in sd:
        if k.endswith("linear_position_embedding"):
            continue
        assert k in P, k
        assert tuple(sd[k].shape) == tuple(P[k].shape), (k, sd[k].shape, P[k].shape)
    missing = [k for k in P if k not in sd]
    assert not missing, missing
    load = {k: (P[k] if k in P else sd[k]) for k in sd}
    model._module.load_state_dict(load)
    # parameter order of the optimiser groups must equal param_shapes order
    spec = O.param_shapes(cfg)
    id2name = {id(p): n for n, p in model._module.named_parameters()}
    for g in range(3):
        ref_names = [id2name[id(p)] for p in model._opt.param_groups[g]["params"]]
        my_names = [n for n, _, gg, _ in spec if gg == g]
        assert ref_names == my_names, (g, ref_names[:5], my_names[:5])
    # pos grid
    grid_ref = model._module._enc_pos.linear_position_embedding
    assert torch.equal(grid_ref, O.position_grid(cfg.obs_size)), "position grid mismatch"

    g = torch.Generator().manual_seed(seed + 1000)
    obs = torch.rand(B, cfg.obs_channels, cfg.obs_size, cfg.obs_size, generator=g)
    trainer = O.OracleTrainer(cfg, P)
    out = {"B": B, "seed": seed, "train_dropout": int(train_dropout)}
    orig_dropout = F.dropout
    for step in range(n_steps):
        noise = O.make_noise(cfg, B, seed + step)
        masks = O.make_masks(cfg, B, seed + 77 + step) if train_dropout else None
        # ---- reference
        if train_dropout:
            model.train()
            torch.nn.functional.dropout = DropoutReplay([masks[k] for k in mask_order(cfg)], cfg.dropout)
        else:
            model.eval()
        try:
            torch.manual_seed(seed + step)      # reproduces make_noise's three draws
            m_ref = model.update(obs, None, step)
        finally:
            torch.nn.functional.dropout = orig_dropout
        # ---- oracle
        res = trainer.update(obs, noise, step, masks)
        # ---- compare
        def chk(name, a, b, tol=2e-5):
            a = torch.as_tensor(a).double()
            b = torch.as_tensor(b).double()
            err = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)
            assert err < tol, f"{tag} step {step} {name}: rel err {err}"
            return err
        chk("loss", res["loss"].detach(), m_ref["loss"].detach())
        chk("dvae_mse", res["dvae_mse"].detach(), m_ref["dvae_mse"])
        chk("ce", res["cross_entropy"].detach(), m_ref["cross_entropy"])
        chk("norm", res["norm"], m_ref["norm"])
        assert abs(res["tau"] - m_ref["tau"].item()) < 1e-6
        worst = 0.0
        for n, p in model._module.named_parameters():
            if not p.requires_grad:
                continue
            worst = max(worst, chk("param " + n, trainer.P[n].detach(), p.detach(), 5e-5))
        print(f"[{tag}] step {step}: loss {m_ref['loss'].item():.6f} norm {float(m_ref['norm']):.6f} "
              f"max param rel err {worst:.2e}")
        out[f"s{step}.loss"] = np.float64(m_ref["loss"].item())
        out[f"s{step}.dvae_mse"] = np.float64(m_ref["dvae_mse"].item())
        out[f"s{step}.cross_entropy"] = np.float64(m_ref["cross_entropy"].item())
        out[f"s{step}.norm"] = np.float64(float(m_ref["norm"]))
        out[f"s{step}.tau"] = np.float64(m_ref["tau"].item())
        for k in ("lr_dvae", "lr_enc", "lr_dec"):
            out[f"s{step}.{k}"] = np.float64(m_ref[k].item())

    # parameters after n_steps updates: checksums for every tensor (+ full tensors for the tiny case)
    names, sums = [], []
    for n, p in model._module.named_parameters():
        if not p.requires_grad:
            continue
        names.append(n)
        sums.append(summarize(p))
        out["paramhead." + n] = p.detach().flatten()[:16].numpy().copy()
    out["param_names"] = np.array(names)
    out["param_sums"] = np.stack(sums)

    # one more forward (no update) for intermediates + raw gradients
    step = n_steps
    noise = O.make_noise(cfg, B, seed + step)
    model.eval()
    model._opt.zero_grad()
    model._module.update_tau(step)
    torch.manual_seed(seed + step)
    mod = model._module
    z, z_hard = mod._get_z(obs)
    recon = mod._dvae.decode(z)
    torch.manual_seed(seed + step)
    _ = torch.empty_like(noise["z"]).exponential_()
    _ = torch.empty_like(noise["z"]).exponential_()
    slots, attns, ce = mod._get_slots(obs, z_hard=z_hard, with_attns=True, with_ce=True)
    dvae_mse = ((obs - recon) ** 2).sum() / B
    (dvae_mse + ce).backward()
    tokens = z_hard.permute(0, 2, 3, 1).flatten(1, 2).argmax(-1)
    trainer2 = O.OracleTrainer(cfg, {n: trainer.P[n].detach() for n in trainer.P})
    res = trainer2.loss_and_grads(obs, noise, step, None)
    rel = lambda a, b: ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-12)).item()
    assert torch.equal(res["tokens"], tokens), "token mismatch"
    assert rel(res["slots"].detach(), slots.detach()) < 2e-5
    assert rel(res["attn"].detach(), attns.detach()) < 2e-5
    assert rel(res["recon"].detach(), recon.detach()) < 2e-5
    gw = 0.0
    gnames, gsums = [], []
    gmax = max(p.grad.abs().max().item() for p in mod.parameters() if p.requires_grad)
    for n, p in mod.named_parameters():
        if not p.requires_grad:
            continue
        # norm_slots.bias has an exactly-zero true gradient (softmax shift invariance): floor the
        # denominator at 1e-6 x the largest gradient so rounding noise is not compared to itself
        e_ = ((trainer2.P[n].grad.double() - p.grad.double()).abs().max() /
              max(p.grad.double().abs().max().item(), 1e-6 * gmax)).item()
        if e_ > grad_tol:
            print("   grad mismatch", n, e_, p.grad.abs().max().item())
        gw = max(gw, e_)
        gnames.append(n)
        gsums.append(summarize(p.grad))
        if full:
            out["grad." + n] = p.grad.numpy().copy()
    assert gw < grad_tol, gw
    out["grad_oracle_vs_reference"] = np.float64(gw)
    print(f"[{tag}] fwd/bwd at step {step}: max grad rel err (max-norm per tensor) {gw:.2e}")
    out["fwd.dvae_mse"] = np.float64(dvae_mse.item())
    out["fwd.cross_entropy"] = np.float64(ce.item())
    out["fwd.slots"] = slots.detach().numpy().copy()
    out["fwd.tokens"] = tokens.numpy().astype(np.int32)
    out["fwd.attn_sums"] = attns.detach().sum(1).numpy().copy()            # [B,K]
    out["fwd.attn_head"] = attns.detach()[:, :64].numpy().copy()           # first 64 positions
    out["fwd.recon_sums"] = summarize(recon)
    out["fwd.recon_head"] = recon.detach()[:, :, :4, :8].numpy().copy()
    out["grad_names"] = np.array(gnames)
    out["grad_sums"] = np.stack(gsums)
    np.savez_compressed(os.path.join(HERE, f"slate_{tag}.npz"), **out)
    print(f"[{tag}] wrote slate_{tag}.npz")


def run_bcdec(SLATE):
    from oracle import slate_oracle as O
    cfg = O.default_cfg(obs_size=16, vocab_size=128, d_model=64, slot_size=64, mlp_hidden=64,
                        num_slots=3, num_iterations=2, num_dec_blocks=1, use_bcdec=True)
    B, seed = 2, 11
    torch.manual_seed(0)
    ocr, env = ref_config(cfg)
    model = SLATE(ocr, env)
    P = O.formula_params(cfg)
    sd = model._module.state_dict()
    model._module.load_state_dict({k: (P[k] if k in P else sd[k]) for k in sd})
    g = torch.Generator().manual_seed(seed + 1000)
    obs = torch.rand(B, 3, 16, 16, generator=g)
    model.eval()
    trainer = O.OracleTrainer(cfg, P)
    out = {}
    for step in range(2):
        noise = O.make_noise(cfg, B, seed + step)
        torch.manual_seed(seed + step)
        m_ref = model.update(obs, None, step)
        res = trainer.update(obs, noise, step, None)
        e = abs(res["loss"].item() - m_ref["loss"].item()) / abs(m_ref["loss"].item())
        assert e < 2e-5, e
        out[f"s{step}.loss"] = np.float64(m_ref["loss"].item())
        out[f"s{step}.norm"] = np.float64(float(m_ref["norm"]))
    names, sums = [], []
    for n, p in model._module.named_parameters():
        if p.requires_grad:
            names.append(n)
            sums.append(summarize(p))
            a, b = trainer.P[n].detach().double(), p.detach().double()
            assert ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item() < 5e-5, n
    out["param_names"] = np.array(names)
    out["param_sums"] = np.stack(sums)
    np.savez_compressed(os.path.join(HERE, "slate_bcdec_tiny.npz"), **out)
    print("[bcdec_tiny] ok, loss", out["s0.loss"], out["s1.loss"])


def main():
    from oracle import slate_oracle as O
    SLATE = import_reference()
    torch.set_num_threads(8)
    tiny = O.default_cfg(obs_size=16, vocab_size=128, d_model=64, slot_size=64, mlp_hidden=64,
                         num_slots=3, num_iterations=2, num_dec_blocks=2)
    # several slot-attention heads (ocrs/common/slot_attn.py:54-92) at shapes the HIP engine runs too: 6 slots x 2 heads (head width 96)
    # and 4 slots x 4 heads (48); one update() and the next forward / backward, as the a64 case
    heads2 = O.default_cfg(obs_size=16, vocab_size=256, num_slots=6, num_iterations=3, num_dec_blocks=2, num_slot_heads=2)
    heads4 = O.default_cfg(obs_size=16, vocab_size=256, num_slots=4, num_iterations=2, num_dec_blocks=1, num_slot_heads=4)
    if "--only-heads" in sys.argv:          # the other fixtures stay as committed
        run_case("heads2_eval", heads2, B=2, seed=11, train_dropout=False, n_steps=1, SLATE=SLATE, full=False)
        run_case("heads4_eval", heads4, B=3, seed=13, train_dropout=False, n_steps=1, SLATE=SLATE, full=False)
        return
    run_case("tiny_eval", tiny, B=2, seed=3, train_dropout=False, n_steps=2, SLATE=SLATE, full=True)
    run_case("tiny_train", tiny, B=2, seed=5, train_dropout=True, n_steps=2, SLATE=SLATE, full=False)
    a64 = O.default_cfg(obs_size=64, num_slots=6)
    run_case("a64_eval", a64, B=2, seed=7, train_dropout=False, n_steps=1, SLATE=SLATE, full=False)
    run_bcdec(SLATE)
    run_case("heads2_eval", heads2, B=2, seed=11, train_dropout=False, n_steps=1, SLATE=SLATE, full=False)
    run_case("heads4_eval", heads4, B=3, seed=13, train_dropout=False, n_steps=1, SLATE=SLATE, full=False)


if __name__ == "__main__":
    main()
