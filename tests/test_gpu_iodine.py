"""IODINE on the GPU through the C ABI against the oracle (oracle/iodine_oracle.py, pinned to the
reference by tests/golden/iodine_*.npz): forward losses and tensors, every parameter gradient, update() steps, the
Python drop-in surface.  Tolerances: losses 1e-5 rel, tensors 1e-4, gradients 1e-3 of each tensor's max, as for SLATE."""
from types import SimpleNamespace as NS

import pytest
import torch

from tests.gpu_util import load_params, log, relerr
from oracle import iodine_oracle as IO

pytestmark = pytest.mark.gpu

TINY = dict(obs_size=16, num_slots=3, num_iterations=3)
S32 = dict(obs_size=32, num_slots=7, num_iterations=5)          # BASELINE config 4's slot / iteration counts
NOLN = dict(obs_size=16, num_slots=4, num_iterations=2, layer_norm=False)
FULL = dict(obs_size=64, num_slots=7, num_iterations=5)          # BASELINE config 4 at full resolution (B = 1 keeps the CPU oracle to seconds)


def dims(cfg):
    return NS(obs_size=cfg.obs_size, obs_channels=3, slot_size=cfg.slot_size, num_iterations=cfg.num_iterations, num_slots=cfg.num_slots,
              sigma=cfg.sigma, beta=cfg.beta, layer_norm=cfg.layer_norm, ref_mlp_hidden=cfg.ref_mlp_hidden)


def make_engine(cfg, B):
    from ocrl_amd.engine import IodineEngine
    return IodineEngine(dims(cfg), max_batch=B)


@pytest.mark.parametrize("tag,over,B", [("io_tiny", TINY, 2), ("io_s32", S32, 2), ("io_noln", NOLN, 3), ("io_full", FULL, 1)])
def test_iodine_forward_backward(tag, over, B):
    cfg = IO.default_cfg(**over)
    P = IO.formula_params(cfg)
    eng = make_engine(cfg, B)
    load_params(eng, P)
    g = torch.Generator().manual_seed(100)
    obs = torch.rand(B, 3, cfg.obs_size, cfg.obs_size, generator=g)
    eps = IO.make_noise(cfg, B, 7)
    tr = IO.OracleTrainer(cfg, P)
    Pg = {k: (v.clone().requires_grad_(True) if k in tr.trainable else v) for k, v in tr.P.items()}
    ref = IO.iodine_forward(Pg, obs, eps, cfg, return_all=True)
    m = eng.forward(obs.cuda(), seed=1, noise=eps.cuda())
    torch.cuda.synchronize()
    K, S, L = cfg.num_slots, cfg.obs_size, cfg.slot_size
    errs = {
        "loss": relerr(m[0], ref["loss"]), "mse": relerr(m[1], ref["mse"]), "kl": relerr(m[2], ref["kl"]),
        "slots": relerr(eng.tensor("slots", (B, K, L)), ref["slots"]),
        "masks": relerr(eng.tensor("masks", (B, K, 1, S, S)), ref["masks"]),
        "recon": relerr(eng.tensor("recon", (B, 3, S, S)), ref["recon"]),
        "recons_masked": relerr(eng.tensor("recons_masked", (B, K, 3, S, S)), ref["recons_masked"]),
        "enc0": relerr(eng.tensor("enc0", (B, K, S, S, 17)).permute(0, 1, 4, 2, 3), ref["trace"][0]["enc"]),
        "latent0": relerr(eng.tensor("xin0", (B * K, cfg.ref_mlp_hidden + 4 * L))[:, cfg.ref_mlp_hidden:], ref["trace"][0]["latent"].reshape(B * K, -1)),
    }
    log(f"[{tag}] forward: " + " ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    for k in ("loss", "mse", "kl"):
        assert errs[k] < 1e-5, (k, errs[k])
    for k in ("slots", "masks", "recon", "recons_masked", "enc0", "latent0"):
        assert errs[k] < 1e-4, (k, errs[k])
    names = list(tr.trainable)
    gs = torch.autograd.grad(ref["loss"], [Pg[n] for n in names])
    eng.backward()
    torch.cuda.synchronize()
    gmax = max(float(x.abs().max()) for x in gs)
    rows = sorted(((relerr(eng.grad(n), x, floor=1e-6 * gmax), n) for n, x in zip(names, gs)), reverse=True)
    log(f"[{tag}] grads: worst {rows[0][0]:.2e}; top: " + "; ".join(f"{n}={e:.1e}" for e, n in rows[:6]))
    assert rows[0][0] < 1e-3, rows[:5]
    assert float(eng.grad("slot_init").abs().max()) == 0.0


def test_iodine_update_steps_match_oracle():
    cfg = IO.default_cfg(**TINY)
    B = 2
    P = IO.formula_params(cfg)
    eng = make_engine(cfg, B)
    load_params(eng, P)
    g = torch.Generator().manual_seed(5)
    obs = torch.rand(B, 3, cfg.obs_size, cfg.obs_size, generator=g)
    tr = IO.OracleTrainer(cfg, P)
    for step in range(3):
        eps = IO.make_noise(cfg, B, 50 + step)
        ref = tr.update(obs, eps)
        m = eng.forward(obs.cuda(), seed=step, noise=eps.cuda())
        eng.backward()
        eng.clip_adam(cfg.lr, cfg.clip)
        torch.cuda.synchronize()
        e_loss, e_norm = relerr(m[0], ref["loss"]), relerr(m[3], ref["norm"])
        log(f"[io_update] step {step}: loss err {e_loss:.2e} norm err {e_norm:.2e}")
        assert e_loss < 2e-5 and e_norm < 1e-4
    worst = 0.0
    for n in tr.trainable:
        d = (eng.param(n).cpu().double() - tr.P[n].double()).abs().max().item()
        # Adam steps of elements whose gradient is rounding noise move by a noise-dependent fraction of lr (see make_golden_iodine.py)
        assert d < max(1e-4 * tr.P[n].abs().max().item(), 0.25 * cfg.lr), (n, d)
        worst = max(worst, d)
    log(f"[io_update] params after 3 steps: worst abs diff {worst:.2e}")
    assert torch.equal(eng.param("slot_init").cpu(), P["slot_init"])


def test_iodine_python_surface():
    """ocrs.Iodine: construction from the reference's config keys, update(), __call__, get_samples, state_dict names"""
    from ocrl_amd import ocrs
    cfg = IO.default_cfg(**TINY)
    ocr = NS(name="Iodine", slot_size=cfg.slot_size, num_iterations=cfg.num_iterations, num_slots=cfg.num_slots, img_channels=3, sigma=cfg.sigma,
             beta=cfg.beta, layer_norm=True, ref_cnn_hidden_size=64, ref_mlp_hidden_size=256, ref_cnn_layers=4, ref_cnn_kernel_size=3,
             ref_cnn_stride_size=2, dec_cnn_hidden_size=64, dec_cnn_layers=4, dec_cnn_kernel_size=3, learning=NS(lr=3e-4, clip=5.0, clip_norm_type=2.0))
    torch.manual_seed(0)
    model = ocrs.Iodine(ocr, NS(obs_size=cfg.obs_size, obs_channels=3))
    assert list(model._module.state_dict().keys()) == [n for n, _, _ in IO.param_shapes(cfg)]
    model.to("cuda:0")
    model.train()
    B, K, S = 4, cfg.num_slots, cfg.obs_size
    obs = torch.rand(B, 3, S, S, device="cuda")
    ids = torch.randint(0, K + 1, (B, S, S), device="cuda")
    masks = torch.nn.functional.one_hot(ids, K + 1).permute(0, 3, 1, 2)[:, :, None].float()
    l0 = None
    for step in range(6):
        met = model.update(obs, masks, step)
        assert set(met) == {"loss", "mse", "ari", "kld", "norm"}
        l0 = l0 if l0 is not None else float(met["loss"])
    assert float(met["loss"]) < l0, (l0, float(met["loss"]))
    with pytest.raises(TypeError):
        model.get_loss(obs, None)                      # the reference requires masks (iodine_module.py:263)
    slots, mk = model(obs, with_masks=True)
    assert slots.shape == (B, K, cfg.slot_size) and mk.shape == (B, K, 1, S, S)
    assert torch.allclose(mk.sum(1), torch.ones_like(mk.sum(1)), atol=1e-5)
    viz = model.get_samples(obs)["samples"]
    assert viz.shape == (B, S, S * (2 + 3 * K), 3) and viz.dtype.name == "uint8"
    ck = model.save()
    model2 = ocrs.Iodine(ocr, NS(obs_size=cfg.obs_size, obs_channels=3))
    model2.to("cuda:0")
    model2.load(ck)
    for (n, a), (_, b) in zip(model._module.state_dict().items(), model2._module.state_dict().items()):
        assert torch.equal(a, b), n
