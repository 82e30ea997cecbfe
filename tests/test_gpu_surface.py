"""The wrapper surface (ocrs.SLATE: __call__ variants, get_loss with ground-truth masks / with_mse, _gen_imgs, load before .to) on the
HIP backend against fixtures produced by the reference's own modules (tests/golden/make_golden_extras.py):
slate_module.py:163-196 (forward variants, _gen_imgs), :207-237 (masks / ARI / with_mse), ocrs/base.py:83-88 (load)."""
import os

import numpy as np
import pytest
import torch

from oracle import slate_oracle as O
from tests.gpu_util import build_wrapper, log, relerr
from tests.test_gpu_slate import dev_noise

pytestmark = pytest.mark.gpu

SURF = dict(obs_size=16, vocab_size=256, num_slots=6, num_iterations=3, num_dec_blocks=2)
BCM = dict(obs_size=16, vocab_size=256, num_slots=5, num_iterations=3, num_dec_blocks=1, use_bcdec=True)


def _seeded_masks(B, K1, S, seed):
    g = torch.Generator().manual_seed(seed)
    lab = torch.randint(0, K1, (B, S // 4, S // 4), generator=g)
    lab = lab.repeat_interleave(4, 1).repeat_interleave(4, 2)
    return torch.nn.functional.one_hot(lab, K1).permute(0, 3, 1, 2).unsqueeze(2).float()


def test_forward_variants_match_reference(golden_dir):
    """model(obs), with_masks, with_attns (whitened maps) and use_cnn_feat ([B,N,C+3] feature map) — slate_module.py:181-196"""
    fx = np.load(os.path.join(golden_dir, "slate_surface.npz"))
    cfg = O.default_cfg(**SURF)
    B, seed = int(fx["B"]), int(fx["seed"])
    P = O.formula_params(cfg)
    obs = torch.rand(B, 3, 16, 16, generator=torch.Generator().manual_seed(seed + 1000)).cuda()
    noise = O.make_noise(cfg, B, seed)
    model = build_wrapper(cfg, P)
    sl = dict(slots=noise["slots"].cuda())
    model._module.inject_noise(sl); slots = model(obs)
    model._module.inject_noise(sl); s_m, masks = model(obs, with_masks=True)
    model._module.inject_noise(sl); s_a, attns = model(obs, with_attns=True)
    assert slots.shape == (B, 6, 192) and masks.shape == (B, 6, 1, 16, 16) and attns.shape == (B, 6, 3, 16, 16)
    assert torch.equal(slots, s_m) and torch.equal(slots, s_a)
    e = dict(slots=relerr(slots, torch.from_numpy(fx["slots"])), with_masks=relerr(masks, torch.from_numpy(fx["with_masks"])),
             with_attns=relerr(attns, torch.from_numpy(fx["with_attns"])))
    mf = build_wrapper(cfg, P, use_cnn_feat=True)
    assert mf.rep_dim == 67 and mf.num_slots == 256
    f = mf(obs)
    assert f.shape == (B, 256, 67)
    e["cnn_feat"] = relerr(f, torch.from_numpy(fx["cnn_feat"]))
    log("[surface] forward variants vs the reference: " + " ".join(f"{k}={v:.2e}" for k, v in e.items()))
    assert max(e.values()) < 1e-4, e


def test_gen_imgs_matches_reference(golden_dir):
    """_gen_imgs (slate_module.py:163-179) and get_loss(with_mse=True) (:234-237): tokens bit-exact, image and mse to 1e-4 / 1e-5"""
    fx = np.load(os.path.join(golden_dir, "slate_surface.npz"))
    cfg = O.default_cfg(**SURF)
    B, seed = int(fx["B"]), int(fx["seed"])
    P = O.formula_params(cfg)
    obs = torch.rand(B, 3, 16, 16, generator=torch.Generator().manual_seed(seed + 1000)).cuda()
    model = build_wrapper(cfg, P)
    mod = model._module
    mod.update_tau(0)
    mod.inject_noise(dev_noise(cfg, O.make_noise(cfg, B, seed)))
    m = mod.get_loss(obs, None, with_mse=True)
    torch.cuda.synchronize()
    toks = mod.engine.tensor("tokens", (B, 16), torch.int32).cpu().numpy()
    assert np.array_equal(toks, fx["gen_tokens"]), (toks, fx["gen_tokens"])
    img = mod.engine.tensor("recon", (B, 16, 16, 4))[..., :3].permute(0, 3, 1, 2)
    e_img = relerr(img, torch.from_numpy(fx["gen_image"]))
    e_mse = abs(float(m["mse"]) - float(fx["with_mse.mse"])) / float(fx["with_mse.mse"])
    e_loss = abs(float(m["loss"]) - float(fx["with_mse.loss"])) / float(fx["with_mse.loss"])
    log(f"[surface] _gen_imgs vs the reference: tokens exact, image {e_img:.2e}, mse {e_mse:.2e}, loss {e_loss:.2e}")
    assert e_img < 1e-4 and e_mse < 1e-5 and e_loss < 1e-5


def test_get_loss_with_ground_truth_masks_matches_reference(golden_dir):
    """slate_module.py:207-225 + utils/tools.py:309-320: foreground-masked attention maps, ARI per image (reported only with use_bcdec)"""
    fx = np.load(os.path.join(golden_dir, "slate_masks_bcdec.npz"))
    cfg = O.default_cfg(**BCM)
    B, seed = int(fx["B"]), int(fx["seed"])
    obs = torch.rand(B, 3, 16, 16, generator=torch.Generator().manual_seed(seed + 1000)).cuda()
    masks = _seeded_masks(B, cfg.num_slots + 1, 16, seed).cuda()
    noise = O.make_noise(cfg, B, seed)
    model = build_wrapper(cfg, O.formula_params(cfg))
    model._module.inject_noise(dict(slots=noise["slots"].cuda()))
    m = model._module.get_loss(obs, masks)
    e_mse = abs(float(m["mse"]) - float(fx["mse"])) / float(fx["mse"])
    e_ari = abs(float(m["ari"]) - float(fx["ari"]))
    log(f"[surface] use_bcdec get_loss(obs, masks): mse {e_mse:.2e}, ari {float(m['ari']):.6f} vs reference {float(fx['ari']):.6f}")
    assert e_mse < 1e-5 and float(m["loss"]) == float(m["mse"]) and e_ari < 1e-6
    # SLATE branch: the same mask code runs, ari is not reported (slate_module.py:231)
    cfg2 = O.default_cfg(**dict(BCM, use_bcdec=False))
    model2 = build_wrapper(cfg2, O.formula_params(cfg2))
    model2._module.update_tau(0)
    model2._module.inject_noise(dev_noise(cfg2, noise))
    m2 = model2._module.get_loss(obs, masks)
    assert sorted(m2.keys()) == [str(k) for k in fx["slate.keys"]]
    assert abs(float(m2["loss"]) - float(fx["slate.loss"])) / float(fx["slate.loss"]) < 1e-5


def test_metrics_are_fresh_tensors():
    """the reference returns new tensors per call; a caller that collects metric dicts and averages later (train_ocr.py:75-87) must
    not see the last batch's values in every entry"""
    cfg = O.default_cfg(**SURF)
    model = build_wrapper(cfg, O.formula_params(cfg))
    a = torch.rand(2, 3, 16, 16, device="cuda")
    m1 = model.get_loss(a, None)
    l1 = float(m1["loss"])
    m2 = model.get_loss(torch.rand(2, 3, 16, 16, device="cuda"), None)
    assert float(m1["loss"]) == l1 and float(m2["loss"]) != l1


def test_load_before_to_device_and_extractor_checkpoint(tmp_path):
    """ocrs/base.py:83-88 callers (sb3s/ocr_extractor.py:33-36, poolings/base.py:24-29) load a pre-training checkpoint BEFORE .to(device):
    weights and the Adam state must arrive in the flat device buffers"""
    from types import SimpleNamespace as NS
    from ocrl_amd import ocrs
    from ocrl_amd.sb3s.ocr_extractor import OCRExtractor
    from tests.gpu_util import reference_style_config
    cfg = O.default_cfg(**SURF)
    ocr, env = reference_style_config(cfg)
    a = ocrs.SLATE(ocr, env); a.to("cuda:0"); a.train()
    obs = torch.rand(2, 3, 16, 16, device="cuda")
    for s in range(2):
        a.update(obs, None, s)
    ck = {k: v for k, v in a.save().items()}
    path = str(tmp_path / "model_latest.pth")
    torch.save({"step": 2, "epoch": 0, "best_val_loss": 1.0, **ck}, path)
    b = ocrs.SLATE(ocr, env)
    b.load(torch.load(path, map_location="cpu", weights_only=True))      # engine does not exist yet
    b.to("cuda:0")
    ea, eb = a._module.engine, b._module.engine
    assert torch.equal(ea.flat_p, eb.flat_p) and torch.equal(ea.flat_m, eb.flat_m) and torch.equal(ea.flat_v, eb.flat_v) and eb.adam_step == 2
    conf = NS(ocr=ocr, env=env, num_envs=2, device="cuda:0",
              pooling=NS(name="Transformer", d_model=128, nhead=8, num_layers=1, pos_emb="None", learn_aux_loss=False, learn_downstream_loss=False,
                         ocr_checkpoint=NS(local_file=path, run_id="")))
    ex = OCRExtractor(None, conf)
    ex._pooling.to("cuda:0")
    assert torch.equal(ex._ocr._module.engine.flat_p, ea.flat_p)
    out = ex(obs)
    assert out.shape == (2, 128) and torch.isfinite(out).all()
