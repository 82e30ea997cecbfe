"""BASELINE config 2 at its own size: Slot-Attention (ocr.use_bcdec=True: CNN encoder + slot attention + spatial-broadcast decoder,
ocrs/common/models.py:110-141, ocrs/slate/slate_module.py:218-225) at 128x128 / 6 slots / 3 iterations.  At S = 128 the broadcast
decoder's first-layer shortcut runs all 25 border classes over 128-wide rows and the 64 -> 4 head its full tile loops — none of which the
16x16 / 32x32 cases of tests/test_gpu_slate.py reach.

  * against the CPU oracle (B = 1): loss, slots, attention, features, reconstruction; every gradient under fixed ReLU decisions;
  * against the reference itself: tests/golden/slate_sa128_eval.npz (one reference update() + the next step's forward and gradient
    checksums, written by tests/golden/make_golden_extras.py from the reference modules);
  * batch additivity of loss and gradients at B = 2."""
import os

import numpy as np
import pytest
import torch

from oracle import slate_oracle as O
from tests.gpu_util import dims_from_cfg, load_params, log, relerr
from tests.test_gpu_slate import GRAD_TOL, compare_grads

pytestmark = pytest.mark.gpu
SA128 = dict(obs_size=128, num_slots=6, num_iterations=3, use_bcdec=True)


def _summ(t):
    t = t.detach().double().flatten().cpu()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


def test_sa128_against_oracle():
    from ocrl_amd.engine import SlateEngine
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cfg = O.default_cfg(**SA128)
    B = 1
    S, N, K, D = cfg.obs_size, cfg.obs_size ** 2, cfg.num_slots, cfg.slot_size
    P = O.formula_params(cfg)
    obs = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(31))
    noise = O.make_noise(cfg, B, 32)
    step = 25
    tau, _ = O.schedules(cfg, step)
    tr = O.OracleTrainer(cfg, P)
    res = tr.loss_and_grads(obs, noise, step, None)
    eng = SlateEngine(dims_from_cfg(cfg), max_batch=B)
    load_params(eng, P)
    eng.forward(obs.cuda(), tau, train=False, seed=1, noise=dict(slots=noise["slots"].cuda()))
    torch.cuda.synchronize()
    m = eng.metrics.cpu()
    errs = dict(loss=abs(m[2].item() - res["loss"].item()) / abs(res["loss"].item()),
                feats=relerr(eng.tensor("feats", (B, N, 64)), res["feats"]), slots=relerr(eng.tensor("slots", (B, K, D)), res["slots"]),
                attn=relerr(eng.tensor("attn", (B, N, K)), res["attn"]),
                recon=relerr(eng.tensor("recon", (B, S, S, 4))[..., :3].permute(0, 3, 1, 2), res["recon_bc"]))
    log("[SA 128x128] forward vs oracle: " + " ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    assert errs["loss"] < 1e-5, errs
    for k in ("feats", "slots", "attn", "recon"):
        assert errs[k] < 1e-4, (k, errs[k])
    eng.backward()
    torch.cuda.synchronize()
    worst, rows = compare_grads("SA 128x128", eng, tr, cfg, P, obs, noise, step)
    assert worst < GRAD_TOL, rows[:5]


def test_sa128_reference_fixture_replay(golden_dir):
    """the reference's own update() with use_bcdec=True at 128x128 (B = 1), then its next forward / backward"""
    from ocrl_amd.engine import SlateEngine
    fx = np.load(os.path.join(golden_dir, "slate_sa128_eval.npz"))
    cfg = O.default_cfg(**SA128)
    B, seed = int(fx["B"]), int(fx["seed"])
    S, N, K, D = cfg.obs_size, cfg.obs_size ** 2, cfg.num_slots, cfg.slot_size
    obs = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(seed + 1000)).cuda()
    eng = SlateEngine(dims_from_cfg(cfg), max_batch=B)
    load_params(eng, O.formula_params(cfg))
    tau, lrs = O.schedules(cfg, 0)
    for i, k in enumerate(("lr_dvae", "lr_enc", "lr_dec")):
        assert lrs[i] == pytest.approx(float(fx[f"s0.{k}"]), rel=1e-6)
    eng.forward(obs, tau, train=False, seed=1, noise=dict(slots=O.make_noise(cfg, B, seed)["slots"].cuda()))
    eng.backward()
    eng.clip_adam(lrs, cfg.clip)
    torch.cuda.synchronize()
    m = eng.metrics.cpu()
    e_mse = abs(m[2].item() - float(fx["s0.mse"])) / float(fx["s0.mse"])
    e_norm = abs(m[3].item() - float(fx["s0.norm"])) / float(fx["s0.norm"])
    log(f"[sa128 fixture] reference update() step 0: mse={e_mse:.2e} norm={e_norm:.2e}")
    assert e_mse < 1e-5 and e_norm < 1e-4
    byname = {p.name: p for p in eng.params}
    worst = 0.0
    for n, ref in zip([str(x) for x in fx["param_names"]], fx["param_sums"]):
        got = _summ(eng.view(eng.flat_p, byname[n]))
        e = max(abs(got[1] - ref[1]) / max(ref[1], 1e-12), abs(got[2] - ref[2]) / max(ref[2], 1e-12))
        worst = max(worst, e)
        assert e < 1e-5, (n, got, ref)
        np.testing.assert_allclose(eng.view(eng.flat_p, byname[n]).flatten()[:16].cpu().numpy(), fx["paramhead." + n], rtol=2e-5, atol=1e-7, err_msg=n)
    log(f"[sa128 fixture] parameters after the reference's update(): worst checksum error {worst:.2e} over {len(fx['param_names'])} tensors")
    # ---- next step
    tau, _ = O.schedules(cfg, 1)
    eng.forward(obs, tau, train=False, seed=2, noise=dict(slots=O.make_noise(cfg, B, seed + 1)["slots"].cuda()))
    eng.backward()
    torch.cuda.synchronize()
    m = eng.metrics.cpu()
    assert abs(m[2].item() - float(fx["fwd.mse"])) / float(fx["fwd.mse"]) < 1e-5
    slots = eng.tensor("slots", (B, K, D)).cpu()
    attn = eng.tensor("attn", (B, N, K)).cpu()
    recon = eng.tensor("recon", (B, S, S, 4))[..., :3].permute(0, 3, 1, 2).cpu()
    e = dict(slots=relerr(slots, torch.from_numpy(fx["fwd.slots"])), attn_head=relerr(attn[:, :64], torch.from_numpy(fx["fwd.attn_head"])),
             attn_sums=relerr(attn.sum(1), torch.from_numpy(fx["fwd.attn_sums"])), recon_head=relerr(recon[:, :, :4, :8], torch.from_numpy(fx["fwd.recon_head"])),
             recon_sums=float(np.max(np.abs(_summ(recon) - fx["fwd.recon_sums"])[1:] / np.abs(fx["fwd.recon_sums"])[1:])))
    log("[sa128 fixture] forward at step 1 vs the reference: " + " ".join(f"{k}={v:.2e}" for k, v in e.items()))
    assert e["slots"] < 1e-4 and e["attn_head"] < 1e-4 and e["attn_sums"] < 1e-4 and e["recon_head"] < 1e-4 and e["recon_sums"] < 1e-5, e
    # gradient L2 norms per tensor (a ReLU coin toss moves a tensor's L2 norm far less than its max-norm: the a64 / a128 replays measure <= 1.2e-5)
    tol = 2e-4
    gmax = max(float(np.sqrt(s[2])) for s in fx["grad_sums"])
    worst = 0.0
    for n, ref in zip([str(x) for x in fx["grad_names"]], fx["grad_sums"]):
        got = _summ(eng.view(eng.flat_g, byname[n]))
        ge = abs(np.sqrt(got[2]) - np.sqrt(ref[2])) / max(np.sqrt(ref[2]), 1e-6 * gmax)
        worst = max(worst, ge)
        assert ge <= tol, (n, ge, tol)
    log(f"[sa128 fixture] gradient L2 norms vs the reference: worst {worst:.2e} (tolerance {tol:.1e})")
    # parameters that get no gradient in this mode (dVAE, transformer decoder, slot projection) hold exact zeros
    seen = set(str(x) for x in fx["grad_names"])
    for p in eng.params:
        if p.name not in seen:
            assert float(eng.view(eng.flat_g, p).abs().max()) == 0.0, p.name


def test_sa128_batch_additivity():
    from ocrl_amd.engine import SlateEngine
    cfg = O.default_cfg(**SA128)
    B = 2
    S, K, D = cfg.obs_size, cfg.num_slots, cfg.slot_size
    obs = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(42)).cuda()
    sn = torch.randn(B, K, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7))
    eng = SlateEngine(dims_from_cfg(cfg), max_batch=B)
    load_params(eng, O.formula_params(cfg))

    def run(sl):
        m = eng.forward(obs[sl].contiguous(), 1.0, train=False, seed=1, noise=dict(slots=sn[sl].contiguous()))
        eng.backward()
        torch.cuda.synchronize()
        nb = obs[sl].shape[0]
        return m[:3].cpu().double() * nb, eng.flat_g.cpu().double() * nb

    l_all, g_all = run(slice(0, B))
    l_a, g_a = run(slice(0, 1))
    l_b, g_b = run(slice(1, 2))
    assert torch.isfinite(l_all).all() and torch.isfinite(g_all).all()
    err_l = ((l_a + l_b - l_all).abs() / l_all.abs().clamp_min(1e-12))[[0, 2]].max().item()
    err_g = ((g_a + g_b - g_all).abs().max() / g_all.abs().max()).item()
    log(f"[SA 128x128 additivity] loss {err_l:.2e}, gradients {err_g:.2e}")
    assert err_l < 1e-5 and err_g < 1e-4
