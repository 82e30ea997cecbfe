"""The oracle (oracle/slate_oracle.py) against the golden vectors produced from the real
reference modules by tests/golden/make_golden.py.  CPU only; runs anywhere."""
import os

import numpy as np
import pytest
import torch

from oracle import slate_oracle as O

TINY = dict(obs_size=16, vocab_size=128, d_model=64, slot_size=64, mlp_hidden=64,
            num_slots=3, num_iterations=2, num_dec_blocks=2)


def _summ(t):
    t = t.detach().double().flatten()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


def _replay(cfg, fx, with_masks):
    B, seed = int(fx["B"]), int(fx["seed"])
    g = torch.Generator().manual_seed(seed + 1000)
    obs = torch.rand(B, cfg.obs_channels, cfg.obs_size, cfg.obs_size, generator=g)
    tr = O.OracleTrainer(cfg, O.formula_params(cfg))
    step = 0
    while f"s{step}.loss" in fx:
        noise = O.make_noise(cfg, B, seed + step)
        masks = O.make_masks(cfg, B, seed + 77 + step) if with_masks else None
        res = tr.update(obs, noise, step, masks)
        for k in ("loss", "dvae_mse", "cross_entropy", "norm"):
            assert float(res[k]) == pytest.approx(float(fx[f"s{step}.{k}"]), rel=2e-5), (step, k)
        assert res["tau"] == pytest.approx(float(fx[f"s{step}.tau"]), rel=1e-6)
        for i, k in enumerate(("lr_dvae", "lr_enc", "lr_dec")):
            assert res["lrs"][i] == pytest.approx(float(fx[f"s{step}.{k}"]), rel=1e-6)
        step += 1
    return tr, obs, step


@pytest.mark.parametrize("tag,over,masks", [
    ("tiny_eval", TINY, False),
    ("tiny_train", TINY, True),
    # multi-head slot attention (slot_attn.py:54-92) at shapes the HIP engine replays too (tests/test_gpu_slate.py)
    ("heads2_eval", dict(obs_size=16, vocab_size=256, num_slots=6, num_iterations=3, num_dec_blocks=2, num_slot_heads=2), False),
    ("heads4_eval", dict(obs_size=16, vocab_size=256, num_slots=4, num_iterations=2, num_dec_blocks=1, num_slot_heads=4), False),
])
def test_update_matches_reference(golden_dir, tag, over, masks):
    fx = np.load(os.path.join(golden_dir, f"slate_{tag}.npz"))
    cfg = O.default_cfg(**over)
    tr, obs, step = _replay(cfg, fx, masks)
    names = [str(n) for n in fx["param_names"]]
    for n, ref in zip(names, fx["param_sums"]):
        got = _summ(tr.P[n])
        np.testing.assert_allclose(got[1:], ref[1:], rtol=2e-5, err_msg=n)
        np.testing.assert_allclose(tr.P[n].detach().flatten()[:16].numpy(), fx["paramhead." + n],
                                   rtol=1e-4, atol=1e-7, err_msg=n)
    # forward/backward intermediates at the next step
    B, seed = int(fx["B"]), int(fx["seed"])
    noise = O.make_noise(cfg, B, seed + step)
    tr2 = O.OracleTrainer(cfg, {n: p.detach() for n, p in tr.P.items()})
    res = tr2.loss_and_grads(obs, noise, step, None)
    assert np.array_equal(res["tokens"].numpy().astype(np.int32), fx["fwd.tokens"])
    np.testing.assert_allclose(res["slots"].detach().numpy(), fx["fwd.slots"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(res["attn"].detach()[:, :64].numpy(), fx["fwd.attn_head"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(res["attn"].detach().sum(1).numpy(), fx["fwd.attn_sums"], rtol=1e-4)
    assert float(res["dvae_mse"]) == pytest.approx(float(fx["fwd.dvae_mse"]), rel=2e-5)
    assert float(res["cross_entropy"]) == pytest.approx(float(fx["fwd.cross_entropy"]), rel=2e-5)
    gmax = max(float(np.sqrt(s[2])) for s in fx["grad_sums"])
    for n, ref in zip([str(x) for x in fx["grad_names"]], fx["grad_sums"]):
        got = _summ(tr2.P[n].grad)
        assert abs(np.sqrt(got[2]) - np.sqrt(ref[2])) <= 2e-4 * max(np.sqrt(ref[2]), 1e-6 * gmax), n
        key = "grad." + n
        if key in fx:
            ref_g = fx[key]
            tol = 2e-4 * max(np.abs(ref_g).max(), 1e-6 * gmax)
            assert np.abs(tr2.P[n].grad.numpy() - ref_g).max() <= tol, n


def test_a64_real_config_matches_reference(golden_dir):
    """Real 64x64 / 6 slots / 3 iters / vocab 4096 configuration (BASELINE configs[0] shape), B=2."""
    fx = np.load(os.path.join(golden_dir, "slate_a64_eval.npz"))
    cfg = O.default_cfg(obs_size=64, num_slots=6)
    tr, obs, step = _replay(cfg, fx, False)
    for n, ref in zip([str(x) for x in fx["param_names"]], fx["param_sums"]):
        np.testing.assert_allclose(_summ(tr.P[n])[1:], ref[1:], rtol=2e-5, err_msg=n)


def test_bcdec_matches_reference(golden_dir):
    fx = np.load(os.path.join(golden_dir, "slate_bcdec_tiny.npz"))
    cfg = O.default_cfg(obs_size=16, vocab_size=128, d_model=64, slot_size=64, mlp_hidden=64,
                        num_slots=3, num_iterations=2, num_dec_blocks=1, use_bcdec=True)
    B, seed = 2, 11
    g = torch.Generator().manual_seed(seed + 1000)
    obs = torch.rand(B, 3, 16, 16, generator=g)
    tr = O.OracleTrainer(cfg, O.formula_params(cfg))
    for step in range(2):
        res = tr.update(obs, O.make_noise(cfg, B, seed + step), step, None)
        assert float(res["loss"]) == pytest.approx(float(fx[f"s{step}.loss"]), rel=2e-5)
        assert float(res["norm"]) == pytest.approx(float(fx[f"s{step}.norm"]), rel=2e-5)
    for n, ref in zip([str(x) for x in fx["param_names"]], fx["param_sums"]):
        np.testing.assert_allclose(_summ(tr.P[n])[1:], ref[1:], rtol=2e-5, err_msg=n)


def test_sa128_real_config_matches_reference(golden_dir):
    """BASELINE config 2 at its real size (use_bcdec, 128x128 / 6 slots / 3 iterations), B=1: the reference's update() and the next
    step's forward / gradient checksums (tests/golden/make_golden_extras.py run_sa128)"""
    fx = np.load(os.path.join(golden_dir, "slate_sa128_eval.npz"))
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    cfg = O.default_cfg(obs_size=128, num_slots=6, num_iterations=3, use_bcdec=True)
    B, seed = int(fx["B"]), int(fx["seed"])
    obs = torch.rand(B, 3, cfg.obs_size, cfg.obs_size, generator=torch.Generator().manual_seed(seed + 1000))
    tr = O.OracleTrainer(cfg, O.formula_params(cfg))
    res = tr.update(obs, O.make_noise(cfg, B, seed), 0, None)
    assert float(res["loss"]) == pytest.approx(float(fx["s0.loss"]), rel=2e-5)
    assert float(res["norm"]) == pytest.approx(float(fx["s0.norm"]), rel=2e-5)
    for n, ref in zip([str(x) for x in fx["param_names"]], fx["param_sums"]):
        np.testing.assert_allclose(_summ(tr.P[n])[1:], ref[1:], rtol=2e-5, err_msg=n)
    tr2 = O.OracleTrainer(cfg, {n: p.detach() for n, p in tr.P.items()})
    r2 = tr2.loss_and_grads(obs, O.make_noise(cfg, B, seed + 1), 1, None)
    assert float(r2["loss"]) == pytest.approx(float(fx["fwd.mse"]), rel=2e-5)
    np.testing.assert_allclose(r2["slots"].detach().numpy(), fx["fwd.slots"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(r2["attn"].detach().sum(1).numpy(), fx["fwd.attn_sums"], rtol=1e-4)
    np.testing.assert_allclose(_summ(r2["recon_bc"])[1:], fx["fwd.recon_sums"][1:], rtol=2e-5)


def test_param_counts_match_survey():
    """S=64: 5 388 099 trainable; S=128: 5 535 555; bcdec adds 515 460."""
    def count(cfg):
        return sum(int(np.prod(s)) for _, s, _, tr in O.param_shapes(cfg) if tr)
    assert count(O.default_cfg(obs_size=64)) == 5388099
    assert count(O.default_cfg(obs_size=128)) == 5535555
    assert count(O.default_cfg(obs_size=128, use_bcdec=True)) - count(O.default_cfg(obs_size=128)) == 515460


def test_schedules():
    cfg = O.default_cfg()
    tau, lrs = O.schedules(cfg, 0)
    assert tau == 1.0 and lrs[0] == 3e-4
    assert lrs[1] == pytest.approx(1e-4 / 30000)
    tau, lrs = O.schedules(cfg, 30000)
    assert tau == pytest.approx(0.1)
    tau, _ = O.schedules(cfg, 15000)
    assert tau == pytest.approx(0.55)


# ------------------------------------------------------------------------------------------- IODINE
@pytest.mark.parametrize("tag,over", [
    ("tiny", dict(obs_size=16, num_slots=3, num_iterations=3)),
    ("s32", dict(obs_size=32, num_slots=7, num_iterations=5)),
])
def test_iodine_oracle_matches_reference(golden_dir, tag, over):
    from oracle import iodine_oracle as IO
    fx = np.load(os.path.join(golden_dir, f"iodine_{tag}.npz"))
    cfg = IO.default_cfg(**over)
    B, seed = int(fx["B"]), int(fx["seed"])
    g = torch.Generator().manual_seed(seed + 1000)
    obs = torch.rand(B, 3, cfg.obs_size, cfg.obs_size, generator=g)
    tr = IO.OracleTrainer(cfg, IO.formula_params(cfg))
    step = 0
    while f"s{step}.loss" in fx:
        res = tr.update(obs, IO.make_noise(cfg, B, seed + step))
        for k, r in (("loss", "loss"), ("mse", "mse"), ("kl", "kld"), ("norm", "norm")):
            assert float(res[k]) == pytest.approx(float(fx[f"s{step}.{r}"]), rel=3e-5), (step, k)
        step += 1
    for n, ref in zip([str(n) for n in fx["param_names"]], fx["param_sums"]):
        np.testing.assert_allclose(_summ(tr.P[n])[1:], ref[1:], rtol=5e-5, err_msg=n)
    tr2 = IO.OracleTrainer(cfg, {n: p.detach() for n, p in tr.P.items()})
    res, grads = tr2.loss_and_grads(obs, IO.make_noise(cfg, B, seed + step))
    assert float(res["loss"]) == pytest.approx(float(fx["f.loss"]), rel=3e-5)
    assert float(res["mse"]) == pytest.approx(float(fx["f.mse"]), rel=3e-5)
    assert float(res["kl"]) == pytest.approx(float(fx["f.kl"]), rel=3e-5)
    np.testing.assert_allclose(res["slots"].numpy(), fx["f.slots"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(res["recon"].flatten()[:64].numpy(), fx["f.recon_head"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(res["masks"].flatten()[:64].numpy(), fx["f.masks_head"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(_summ(res["masks"])[1:], fx["f.masks_sum"][1:], rtol=3e-5)
    gmax = max(float(g.abs().max()) for g in grads.values())
    for n, ref in zip([str(n) for n in fx["grad_names"]], fx["grad_sums"]):
        np.testing.assert_allclose(_summ(grads[n])[1:], ref[1:], rtol=1e-4, atol=1e-6 * gmax, err_msg=n)
        np.testing.assert_allclose(grads[n].flatten()[:16].numpy(), fx["gradhead." + n], rtol=2e-4, atol=1e-6 * gmax, err_msg=n)


# ----------------------------------------------------------------------------------------------- pooling
def _grad_summary(t):
    t = t.detach().double().flatten()
    return np.concatenate([np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()]), t[:: max(1, t.numel() // 509)][:509].numpy()])


@pytest.mark.parametrize("tag", ["default", "ape_l2", "default_train"])
def test_pooling_oracle_matches_reference(golden_dir, tag):
    """oracle/pooling_oracle.py against outputs and gradients of the reference's Transformer_Module (tests/golden/make_golden_pooling.py)"""
    from oracle import pooling_oracle as PO
    fx = np.load(os.path.join(golden_dir, f"pooling_{tag}.npz"))
    rep, K, d, nhead, L, ff, has_pos, B = [int(v) for v in fx["cfg"]]
    cfg = PO.default_cfg(rep_dim=rep, num_slots=K, d_model=d, nhead=nhead, num_layers=L, dim_feedforward=ff, pos_emb="ape" if has_pos else "None")
    P = PO.formula_params(cfg)
    p_drop = float(fx["p_drop"][0])
    masks = None
    if p_drop > 0:
        S = K + 1
        shapes = {"drop1": (B, S, d), "ffn": (B, S, ff), "drop2": (B, S, d)}
        masks = {}
        for k in fx.files:
            if k.startswith("m:"):
                shp = shapes[k.split(".")[-1]]
                masks[k[2:]] = torch.from_numpy(np.unpackbits(fx[k])[: int(np.prod(shp))].reshape(shp).astype(np.float32))
    out, g, ds = PO.loss_and_grads(P, torch.from_numpy(fx["slots"]), cfg, torch.from_numpy(fx["cot"]), masks, p_drop)
    assert np.abs(out.numpy() - fx["out"]).max() < 2e-5 * np.abs(fx["out"]).max()
    assert np.abs(ds.numpy() - fx["dslots"]).max() < 2e-4 * np.abs(fx["dslots"]).max()
    gmax = max(np.abs(fx["g:" + n][3:]).max() for n, _ in PO.param_shapes(cfg))
    for n, _ in PO.param_shapes(cfg):
        ref, got = fx["g:" + n], _grad_summary(g[n])
        assert np.abs(got[3:] - ref[3:]).max() < 2e-4 * max(np.abs(ref[3:]).max(), 1e-3 * gmax), n
        assert abs(got[2] - ref[2]) <= 1e-3 * ref[2] + 1e-12, n


@pytest.mark.parametrize("mode", ["cw", "push"])
def test_gt_state_embeddings_match_reference(golden_dir, mode):
    """oracle restatement of the cw_embedding / push_embedding front ends (transformer_module.py:65-111) + the pinned transformer against the
    reference module's output and embedding-parameter gradients (tests/golden/make_golden_pooling.py run_gt_case)"""
    from oracle import pooling_oracle as PO
    fx = np.load(os.path.join(golden_dir, f"pooling_{mode}.npz"))
    cfg = PO.default_cfg(rep_dim=128, num_slots=5, d_model=128)
    P, G = PO.formula_params(cfg), PO.gt_formula_params(mode)
    Gq = {k: v.clone().requires_grad_(True) for k, v in G.items()}
    state, cot = torch.from_numpy(fx["state"]), torch.from_numpy(fx["cot"])
    emb = PO.gt_embed(Gq, state, mode)
    out = PO.forward(P, emb, cfg)
    (out * cot).sum().backward()
    assert np.abs(out.detach().numpy() - fx["out"]).max() < 2e-5 * np.abs(fx["out"]).max()
    gmax = max(np.abs(fx["g:" + n][3:]).max() for n in G)
    for n in G:
        ref, got = fx["g:" + n], _grad_summary(Gq[n].grad)
        assert np.abs(got[3:] - ref[3:]).max() < 2e-4 * max(np.abs(ref[3:]).max(), 1e-3 * gmax), n
