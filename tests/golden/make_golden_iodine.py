"""Generate the golden vectors that pin ``oracle/iodine_oracle.py`` to the reference IODINE.

Runs ONLY in the build container (needs /root/reference; recipe = SURVEY.md Appendix C): imports the reference's own
``ocrs.iodine.iodine.Iodine``, loads closed-form weights, replays the same ``rsample`` noise, runs ``_forward`` /
``update`` and (1) asserts the oracle restatement agrees with the reference, (2) writes small ``.npz`` fixtures under
tests/golden/ for tests/test_oracle_golden.py.  Fixtures are data only.

    python tests/golden/make_golden_iodine.py
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
    from ocrs.iodine.iodine import Iodine  # noqa
    return Iodine


def ref_config(cfg):
    NS = types.SimpleNamespace
    ocr = NS(name="Iodine", slot_size=cfg.slot_size, num_iterations=cfg.num_iterations, num_slots=cfg.num_slots, img_channels=cfg.obs_channels,
             sigma=cfg.sigma, beta=cfg.beta, layer_norm=cfg.layer_norm, ref_cnn_hidden_size=cfg.ref_cnn_hidden,
             ref_mlp_hidden_size=cfg.ref_mlp_hidden, ref_cnn_layers=cfg.ref_cnn_layers, ref_cnn_kernel_size=cfg.ref_cnn_kernel,
             ref_cnn_stride_size=cfg.ref_cnn_stride, dec_cnn_hidden_size=cfg.dec_cnn_hidden, dec_cnn_layers=cfg.dec_cnn_layers,
             dec_cnn_kernel_size=cfg.dec_cnn_kernel, learning=NS(lr=cfg.lr, clip=cfg.clip, clip_norm_type=cfg.clip_norm_type))
    env = NS(obs_size=cfg.obs_size, obs_channels=cfg.obs_channels)
    return ocr, env


class RsampleReplay:
    """Replace Normal.rsample by loc + scale * eps[i] in call order (the reference draws once per iteration)."""

    def __init__(self, eps):
        self.eps, self.i = eps, 0

    def __call__(self, dist, sample_shape=torch.Size()):
        e = self.eps[self.i]
        self.i += 1
        return dist.loc + dist.scale * e


def summarize(t):
    t = t.detach().double().flatten()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


def gt_masks(B, K, S, seed):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, K + 1, (B, S, S), generator=g)
    return torch.nn.functional.one_hot(ids, K + 1).permute(0, 3, 1, 2)[:, :, None].float()      # [B,K+1,1,S,S]


def run_case(tag, cfg, B, seed, n_steps, Iodine):
    from torch.distributions import Normal
    from oracle import iodine_oracle as O

    torch.manual_seed(0)
    ocr, env = ref_config(cfg)
    model = Iodine(ocr, env)
    P = O.formula_params(cfg)
    sd = model._module.state_dict()
    assert list(sd.keys()) == [n for n, _, _ in O.param_shapes(cfg)], (list(sd.keys()), [n for n, _, _ in O.param_shapes(cfg)])
    for k in sd:
        assert tuple(sd[k].shape) == tuple(P[k].shape), (k, sd[k].shape, P[k].shape)
    model._module.load_state_dict(P)
    assert [n for n, _ in model._module.named_parameters()] == [n for n, _, _ in O.param_shapes(cfg)]
    model.train()

    g = torch.Generator().manual_seed(seed + 1000)
    obs = torch.rand(B, 3, cfg.obs_size, cfg.obs_size, generator=g)
    masks = gt_masks(B, cfg.num_slots, cfg.obs_size, seed + 5)
    trainer = O.OracleTrainer(cfg, P)
    out = {"B": B, "seed": seed}
    orig = Normal.rsample

    def chk(name, a, b, tol=2e-5):
        a = torch.as_tensor(a).double()
        b = torch.as_tensor(b).double()
        err = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)
        assert err < tol, f"{tag} {name}: rel err {err}"
        return err

    for step in range(n_steps):
        eps = O.make_noise(cfg, B, seed + step)
        _rp = RsampleReplay(eps)
        Normal.rsample = lambda self, sample_shape=torch.Size(), _rp=_rp: _rp(self)
        try:
            m_ref = model.update(obs, masks, step)
        finally:
            Normal.rsample = orig
        res = trainer.update(obs, eps)
        chk("loss", res["loss"], m_ref["loss"].detach())
        chk("mse", res["mse"], m_ref["mse"])
        chk("kld", res["kl"], m_ref["kld"])
        chk("norm", res["norm"], m_ref["norm"])
        worst = 0.0
        for n, p in model._module.named_parameters():
            if p.grad is None:
                continue
            # Adam's first steps are +-lr * g / (|g| + eps): elements whose gradient is at rounding-noise level move by a
            # noise-dependent fraction of lr, so the bound is max(5e-5 of the tensor, 0.2 lr); the gradients themselves are
            # compared below at 1e-4 and the next step's loss / norm at 2e-5.
            d = (trainer.P[n].double() - p.detach().double()).abs().max().item()
            assert d < max(5e-5 * p.detach().abs().max().item(), 0.2 * cfg.lr), (n, d)
            worst = max(worst, d / p.detach().abs().max().item())
        print(f"[{tag}] step {step}: loss {m_ref['loss'].item():.6f} norm {float(m_ref['norm']):.6f} ari {m_ref['ari']:.4f} max param rel err {worst:.2e}")
        for k in ("loss", "mse", "kld", "norm"):
            out[f"s{step}.{k}"] = np.float64(float(m_ref[k]))
    names, sums = [], []
    for n, p in model._module.named_parameters():
        names.append(n)
        sums.append(summarize(p))
        out["paramhead." + n] = p.detach().flatten()[:16].numpy().copy()
    out["param_names"] = np.array(names)
    out["param_sums"] = np.stack(sums)

    # one more forward + backward (no update): outputs and raw gradients
    step = n_steps
    eps = O.make_noise(cfg, B, seed + step)
    model._opt.zero_grad()
    _rp = RsampleReplay(eps)
    Normal.rsample = lambda self, sample_shape=torch.Size(), _rp=_rp: _rp(self)
    try:
        slots, recon, recons_masked, mk, loss, mse, kl, means, _ = model._module._forward(obs)
    finally:
        Normal.rsample = orig
    loss.backward()
    tr2 = O.OracleTrainer(cfg, {n: trainer.P[n].detach() for n in trainer.P})
    res, grads = tr2.loss_and_grads(obs, eps)
    for k, ref in (("slots", slots), ("recon", recon), ("recons_masked", recons_masked), ("masks", mk), ("loss", loss), ("mse", mse), ("kl", kl), ("means", means)):
        chk("fwd " + k, res[k], ref.detach())
    gmax = max(p.grad.abs().max().item() for p in model._module.parameters() if p.grad is not None)
    gw, gnames, gsums = 0.0, [], []
    for n, p in model._module.named_parameters():
        if p.grad is None:
            assert n == "slot_init", n
            continue
        e = (grads[n].double() - p.grad.double()).abs().max().item() / max(p.grad.abs().max().item(), 1e-6 * gmax)
        assert e < 1e-4, (n, e)
        gw = max(gw, e)
        gnames.append(n)
        gsums.append(summarize(p.grad))
        out["gradhead." + n] = p.grad.flatten()[:16].numpy().copy()
    print(f"[{tag}] forward/backward at step {step}: loss {loss.item():.6f}, worst grad rel err {gw:.2e}")
    out["grad_names"] = np.array(gnames)
    out["grad_sums"] = np.stack(gsums)
    out["f.loss"] = np.float64(loss.item())
    out["f.mse"] = np.float64(mse.item())
    out["f.kl"] = np.float64(kl.item())
    out["f.slots"] = slots.detach().numpy().copy()
    out["f.masks_sum"] = summarize(mk)
    out["f.recon_sum"] = summarize(recon)
    out["f.recon_head"] = recon.detach().flatten()[:64].numpy().copy()
    out["f.masks_head"] = mk.detach().flatten()[:64].numpy().copy()
    np.savez_compressed(os.path.join(HERE, f"iodine_{tag}.npz"), **out)
    print(f"[{tag}] wrote iodine_{tag}.npz")


def main():
    from oracle import iodine_oracle as O
    Iodine = import_reference()
    torch.set_num_threads(8)
    run_case("tiny", O.default_cfg(obs_size=16, num_slots=3, num_iterations=3), B=2, seed=11, n_steps=2, Iodine=Iodine)
    run_case("s32", O.default_cfg(obs_size=32, num_slots=7, num_iterations=5), B=2, seed=23, n_steps=2, Iodine=Iodine)


if __name__ == "__main__":
    main()
