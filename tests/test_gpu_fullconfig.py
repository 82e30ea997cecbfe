"""Parity at the BASELINE configurations' real sizes: the HIP step through the C ABI against
  * the reference-made fixtures tests/golden/slate_a64_eval.npz (64x64) and slate_a128_eval.npz (config A, 128x128) — one reference
    update() (loss terms, norm, every parameter after the step) and the forward of the next step;
  * the CPU oracle at config A (128x128 / 6 slots / 3 iterations / vocab 4096 / 4 blocks, B=2) and config Z (256x256 / 16 slots, B=1),
    every forward stage and every parameter gradient;
  * the batch-additivity property at 256x256 (B=2 vs its halves).

Gradient tolerance.  At these sizes a step evaluates 10^7-10^8 ReLU pre-activations, and a few of them lie inside fp32 rounding noise
of zero; whichever way such a unit's mask falls, one token's / pixel's contribution (~1/T .. 1/N of a weight-gradient row) appears or
vanishes.  Any two fp32 evaluations therefore differ by up to ~1e-2 of a tensor's max on the tensors upstream of a flip: the fp32 oracle
against an fp64 run of itself, the oracle against the reference modules (make_golden_extras.py), and this backend against either
(tests/diag_encoder_grad.py shows the mechanism on the 64x64 case: one flipped mask at |pre-activation| = 4.9e-8 explains all of a
2.7e-4 error, the weight-gradient kernel itself is at 2.8e-7).  A fixed max-norm tolerance is then either vacuous or unmeetable, so the
comparison that is asserted holds the ReLU decisions fixed: the oracle runs in fp64 with the masks of the HIP forward imposed
(tests/gpu_util.py) and every tensor must agree within GRAD_TOL.  The as-is errors of the HIP path and of the fp32 oracle against the
plain fp64 run, and the number of flipped masks on either side, are logged next to it (whether a flip occurs on a given input is a coin
toss for either fp32 evaluation: observed both ways).  Errors are max-norm, relative to the tensor's max (floored at 1e-5 x the largest gradient).  The ReLU masks
of both oracle runs are recorded and the flip counts logged next to the errors."""
import os

import numpy as np
import pytest
import torch

from oracle import slate_oracle as O
from tests.gpu_util import assert_knife_edge, dims_from_cfg, grad_floor, hip_relu_masks, load_params, log, mask_matched_fp64_grads, relerr
from tests.test_gpu_slate import GRAD_TOL, compare_forward, dev_noise

pytestmark = pytest.mark.gpu



def _summ(t):
    t = t.detach().double().flatten().cpu()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


class _ReluRecorder:
    """records the mask of every F.relu call of an oracle run, in call order"""

    def __init__(self):
        self.masks = []
        self._orig = torch.nn.functional.relu

    def __enter__(self):
        def relu(x, inplace=False):
            self.masks.append((x > 0).detach())
            return self._orig(x)
        torch.nn.functional.relu = relu
        return self

    def __exit__(self, *a):
        torch.nn.functional.relu = self._orig


def run_oracles(cfg, P, obs, noise, step):
    """fp32 oracle (the reference's arithmetic) and an fp64 run of the same restatement; returns the number of ReLU masks that differ"""
    t32 = O.OracleTrainer(cfg, P)
    with _ReluRecorder() as r32m:
        r32 = t32.loss_and_grads(obs, noise, step, None)
    P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
    t64 = O.OracleTrainer(cfg, P64)
    with _ReluRecorder() as r64m:
        t64.loss_and_grads(obs.double(), {k: v.double() for k, v in noise.items()}, step, None)
    assert len(r32m.masks) == len(r64m.masks)
    flips = [int((a != b).sum()) for a, b in zip(r32m.masks, r64m.masks)]
    total = sum(int(a.numel()) for a in r64m.masks)
    log(f"   ReLU masks fp32 oracle vs fp64 oracle: {sum(flips)} flips in {total} units; per site (non-zero): " +
        ", ".join(f"#{i}:{f}" for i, f in enumerate(flips) if f))
    return t32, r32, t64


def grade_gradients(tag, eng, t32, t64):
    gmax = max(t64.P[p.name].grad.abs().max().item() for p in eng.params)
    rows = []
    for p in eng.params:
        ref = t64.P[p.name].grad.reshape(p.shape)
        e_hip = relerr(eng.view(eng.flat_g, p), ref, floor=grad_floor(p.name, gmax))
        e_o32 = relerr(t32.P[p.name].grad.reshape(p.shape), ref, floor=grad_floor(p.name, gmax))
        rows.append((e_hip, e_o32, p.name))
    worst_o32 = max(r[1] for r in rows)
    rows.sort(reverse=True)
    med = sorted(r[0] for r in rows)[len(rows) // 2]
    log(f"[{tag}] gradients vs fp64 oracle over {len(rows)} tensors: HIP worst {rows[0][0]:.2e} median {med:.2e}; fp32 oracle worst {worst_o32:.2e} median "
        f"{sorted(r[1] for r in rows)[len(rows) // 2]:.2e}; top (HIP/fp32-oracle): " + "; ".join(f"{n} {a:.1e}/{b:.1e}" for a, b, n in rows[:6]))
    return rows[0][0], worst_o32


def replay_reference_fixture(tag, cfg, fx):
    from ocrl_amd.engine import SlateEngine
    B, seed = int(fx["B"]), int(fx["seed"])
    S, E = cfg.obs_size, cfg.obs_size // 4
    T, N, V, K, D = E * E, S * S, cfg.vocab_size, cfg.num_slots, cfg.slot_size
    obs = torch.rand(B, cfg.obs_channels, S, S, generator=torch.Generator().manual_seed(seed + 1000))
    eng = SlateEngine(dims_from_cfg(cfg), max_batch=B)
    load_params(eng, O.formula_params(cfg))
    # ---- the reference's update() at step 0
    step = 0
    tau, lrs = O.schedules(cfg, step)
    assert tau == pytest.approx(float(fx["s0.tau"]), rel=1e-6)
    for i, k in enumerate(("lr_dvae", "lr_enc", "lr_dec")):
        assert lrs[i] == pytest.approx(float(fx[f"s0.{k}"]), rel=1e-6)
    eng.forward(obs.cuda(), tau, train=False, seed=1, noise=dev_noise(cfg, O.make_noise(cfg, B, seed + step)))
    eng.backward()
    eng.clip_adam(lrs, cfg.clip)
    torch.cuda.synchronize()
    m = eng.metrics.cpu()
    errs = {k: abs(m[i].item() - float(fx[f"s0.{k}"])) / abs(float(fx[f"s0.{k}"])) for i, k in ((0, "dvae_mse"), (1, "cross_entropy"), (2, "loss"), (3, "norm"))}
    log(f"[{tag}] reference update() step 0: " + " ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    assert errs["dvae_mse"] < 1e-5 and errs["cross_entropy"] < 1e-5 and errs["loss"] < 1e-5 and errs["norm"] < 1e-4, errs
    names = [str(n) for n in fx["param_names"]]
    worst = 0.0
    byname = {p.name: p for p in eng.params}
    for n, ref in zip(names, fx["param_sums"]):
        got = _summ(eng.view(eng.flat_p, byname[n]))
        e = max(abs(got[1] - ref[1]) / max(ref[1], 1e-12), abs(got[2] - ref[2]) / max(ref[2], 1e-12))
        worst = max(worst, e)
        assert e < 1e-5, (n, got, ref)                      # sum|p| and sum p^2 of every tensor after the step (weights 1e-5)
        head = eng.view(eng.flat_p, byname[n]).flatten()[:16].cpu().numpy()
        np.testing.assert_allclose(head, fx["paramhead." + n], rtol=2e-5, atol=1e-7, err_msg=n)
    log(f"[{tag}] parameters after the reference's update(): worst checksum error {worst:.2e} over {len(names)} tensors")
    # ---- forward + backward of the next step
    step = 1
    tau, _ = O.schedules(cfg, step)
    eng.forward(obs.cuda(), tau, train=False, seed=2, noise=dev_noise(cfg, O.make_noise(cfg, B, seed + step)))
    eng.backward()
    torch.cuda.synchronize()
    m = eng.metrics.cpu()
    assert abs(m[0].item() - float(fx["fwd.dvae_mse"])) / float(fx["fwd.dvae_mse"]) < 1e-5
    assert abs(m[1].item() - float(fx["fwd.cross_entropy"])) / float(fx["fwd.cross_entropy"]) < 1e-5
    assert np.array_equal(eng.tensor("tokens", (B, T), torch.int32).cpu().numpy(), fx["fwd.tokens"])
    slots = eng.tensor("slots", (B, K, D)).cpu()
    attn = eng.tensor("attn", (B, N, K)).cpu()
    recon = eng.tensor("recon", (B, S, S, 4))[..., :3].permute(0, 3, 1, 2).cpu()
    e = dict(slots=relerr(slots, torch.from_numpy(fx["fwd.slots"])), attn_head=relerr(attn[:, :64], torch.from_numpy(fx["fwd.attn_head"])),
             attn_sums=relerr(attn.sum(1), torch.from_numpy(fx["fwd.attn_sums"])), recon_head=relerr(recon[:, :, :4, :8], torch.from_numpy(fx["fwd.recon_head"])),
             recon_sums=float(np.max(np.abs(_summ(recon) - fx["fwd.recon_sums"])[1:] / np.abs(fx["fwd.recon_sums"])[1:])))
    log(f"[{tag}] forward at step 1 vs the reference: " + " ".join(f"{k}={v:.2e}" for k, v in e.items()))
    assert e["slots"] < 1e-4 and e["attn_head"] < 1e-4 and e["attn_sums"] < 1e-4 and e["recon_head"] < 1e-4 and e["recon_sums"] < 1e-5, e
    # gradient checksums: the L2 norm of every tensor.  The reference's fp32 run takes its own ReLU decisions and a decision at a rounding tie
    # may fall the other way here (fx["grad_oracle_vs_reference"]: the fp32 oracle is up to 1e-2 max-norm from the reference on such a tensor);
    # a tensor's L2 norm moves by far less: measured 3.7e-6 (a128) / 1.2e-5 (a64) without a flip and 2.1e-4 (a128, first encoder
    # convolution) with one.  Tolerance: a tenth of the fixture's max-norm gap, at least 2e-4
    tol = max(2e-4, 0.1 * float(fx["grad_oracle_vs_reference"]))
    gmax = max(float(np.sqrt(s[2])) for s in fx["grad_sums"])
    worst = 0.0
    for n, ref in zip([str(x) for x in fx["grad_names"]], fx["grad_sums"]):
        got = _summ(eng.view(eng.flat_g, byname[n]))
        ge = abs(np.sqrt(got[2]) - np.sqrt(ref[2])) / max(np.sqrt(ref[2]), 1e-6 * gmax)
        worst = max(worst, ge)
        assert ge <= tol, (n, ge, tol)
    log(f"[{tag}] gradient L2 norms vs the reference: worst {worst:.2e} (tolerance {tol:.1e})")


def test_a64_reference_fixture_replay(golden_dir):
    """BASELINE configs[0] shape: 64x64 / 6 slots / 3 iterations / vocab 4096 / 4 blocks, B=2 — against the reference's own numbers"""
    fx = np.load(os.path.join(golden_dir, "slate_a64_eval.npz"))
    replay_reference_fixture("a64 fixture", O.default_cfg(obs_size=64, num_slots=6), fx)


@pytest.mark.parametrize("tag,over", [("heads2", dict(obs_size=16, vocab_size=256, num_slots=6, num_iterations=3, num_dec_blocks=2, num_slot_heads=2)),
                                      ("heads4", dict(obs_size=16, vocab_size=256, num_slots=4, num_iterations=2, num_dec_blocks=1, num_slot_heads=4))])
def test_multi_head_reference_fixture_replay(golden_dir, tag, over):
    """num_slot_heads = 2 / 4 (ocrs/common/slot_attn.py:54-92) — one reference update() and the next forward / backward, against the reference's numbers"""
    fx = np.load(os.path.join(golden_dir, f"slate_{tag}_eval.npz"))
    replay_reference_fixture(f"{tag} fixture", O.default_cfg(**over), fx)


def test_a128_reference_fixture_replay(golden_dir):
    """config A (the headline metric's shape), B=1 — against the reference's own numbers"""
    fx = np.load(os.path.join(golden_dir, "slate_a128_eval.npz"))
    replay_reference_fixture("a128 fixture", O.default_cfg(obs_size=128, num_slots=6), fx)


@pytest.mark.parametrize("tag,over,B", [
    ("config A 128x128/6 slots", dict(obs_size=128, num_slots=6, num_iterations=3), 2),
    ("config Z 256x256/16 slots", dict(obs_size=256, num_slots=16, num_iterations=3), 1),
    ("config A64 64x64/6 slots", dict(obs_size=64, num_slots=6, num_iterations=3), 3),
])
def test_full_config_against_oracle(tag, over, B):
    """every forward stage and every parameter gradient at the configuration's real size"""
    from ocrl_amd.engine import SlateEngine
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cfg = O.default_cfg(**over)
    P = O.formula_params(cfg)
    obs = torch.rand(B, 3, cfg.obs_size, cfg.obs_size, generator=torch.Generator().manual_seed(77))
    noise = O.make_noise(cfg, B, 78)
    step = 25
    tau, _ = O.schedules(cfg, step)
    t32, r32, t64 = run_oracles(cfg, P, obs, noise, step)
    eng = SlateEngine(dims_from_cfg(cfg), max_batch=B)
    load_params(eng, P)
    eng.forward(obs.cuda(), tau, train=False, seed=1, noise=dev_noise(cfg, noise))
    torch.cuda.synchronize()
    errs = compare_forward(tag, eng, cfg, r32, B, noise, tau)
    assert errs["tokens_mismatch"] == 0
    for k in ("dvae_mse", "cross_entropy", "loss"):
        assert errs[k] < 1e-5, (k, errs[k])
    for k in ("z_logits", "z", "recon", "feats", "slots", "attn", "dec_out"):
        assert errs[k] < 1e-4, (k, errs[k])
    eng.backward()
    torch.cuda.synchronize()
    grade_gradients(tag, eng, t32, t64)          # logged, not asserted: with each side's own ReLU decisions the comparison is a coin toss (see above)
    # with the ReLU decisions of the HIP forward held fixed in the fp64 run: tight, per tensor
    t64m, fr = mask_matched_fp64_grads(cfg, P, obs, noise, step, hip_relu_masks(eng, cfg, B))
    assert_knife_edge(fr, tag)
    gmax = max(t64m.P[p.name].grad.abs().max().item() for p in eng.params)
    rows = sorted(((relerr(eng.view(eng.flat_g, p), t64m.P[p.name].grad.reshape(p.shape), floor=grad_floor(p.name, gmax)), p.name) for p in eng.params), reverse=True)
    log(f"[{tag}] gradients vs mask-matched fp64 oracle: worst {rows[0][0]:.2e} ({fr.flips} of {fr.units} ReLU decisions of the HIP forward differ from "
        f"fp64's own" + (f", largest |pre-activation| among them {max(fr.min_flipped):.1e}" if fr.flips else "") + "); top: "
        + "; ".join(f"{n}={e:.1e}" for e, n in rows[:5]))
    assert rows[0][0] < GRAD_TOL, rows[:5]


def test_config_z_batch_additivity():
    """256x256 / 16 slots: B * (loss, gradient) of a batch equals the sum over its halves (no operator mixes images)"""
    from ocrl_amd.engine import SlateEngine
    cfg = O.default_cfg(obs_size=256, num_slots=16, num_iterations=3)
    B = 2
    S, T, V, K, D = 256, 4096, cfg.vocab_size, cfg.num_slots, cfg.slot_size
    obs = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(42)).cuda()
    gg = torch.Generator(device="cuda").manual_seed(7)
    noise = dict(z=torch.empty(B, T, V, device="cuda").exponential_(generator=gg), z_hard=torch.empty(B, T, V, device="cuda").exponential_(generator=gg),
                 slots=torch.randn(B, K, D, device="cuda", generator=gg))
    eng = SlateEngine(dims_from_cfg(cfg), max_batch=B)
    load_params(eng, O.formula_params(cfg))

    def run(sl):
        n = {k: v[sl].contiguous() for k, v in noise.items()}
        m = eng.forward(obs[sl].contiguous(), 0.7, train=False, seed=1, noise=n)
        eng.backward()
        torch.cuda.synchronize()
        nb = obs[sl].shape[0]
        return m[:3].cpu().double() * nb, eng.flat_g.cpu().double() * nb

    l_all, g_all = run(slice(0, B))
    l_a, g_a = run(slice(0, 1))
    l_b, g_b = run(slice(1, 2))
    assert torch.isfinite(l_all).all() and torch.isfinite(g_all).all()
    err_l = ((l_a + l_b - l_all).abs() / l_all.abs().clamp_min(1e-12)).max().item()
    err_g = ((g_a + g_b - g_all).abs().max() / g_all.abs().max()).item()
    log(f"[config Z additivity] loss terms {err_l:.2e}, gradients {err_g:.2e}")
    assert err_l < 1e-5 and err_g < 1e-4
