"""Weight initialisation: per-tensor statistics of ocrs.SLATE's constructor against statistics recorded from the
reference's constructors (ocrs/common/networks.py:6-74, slot_attn.py:133-136, transformer.py:53-58,193-198, slate_module.py:273,287-288;
fixture: tests/golden/slate_init_stats.npz = mean over four seeded constructions).  Statistical, not bitwise, parity: which tensors get
kaiming / xavier / the (3*blocks)^-0.5 gain / orthogonal / zeros / N(0,1) / trunc-normal shows in their standard deviation and range."""
import os

import numpy as np
import torch

from oracle import slate_oracle as O
from tests.gpu_util import reference_style_config


def test_reference_init_statistics(golden_dir):
    from ocrl_amd import ocrs
    fx = np.load(os.path.join(golden_dir, "slate_init_stats.npz"))
    names = [str(n) for n in fx["names"]]
    ref = {n: r for n, r in zip(names, fx["stats"])}
    cfg = O.default_cfg(obs_size=64, num_slots=6)
    runs = []
    for s in range(4):
        torch.manual_seed(500 + s)
        ocr, env = reference_style_config(cfg)
        mod = ocrs.SLATE(ocr, env)._module
        cur = {}
        for n, p in mod.named_parameters():
            if not p.requires_grad:
                continue
            t = p.detach().double()
            orth = (t.T @ t - torch.eye(t.shape[1], dtype=torch.float64)).abs().max().item() if n.endswith("gru.weight_hh") else -1.0
            cur[n] = (t.mean().item(), t.std(unbiased=False).item(), t.abs().max().item(), orth, t.numel())
        runs.append(cur)
    assert sorted(runs[0]) == sorted(names), set(names) ^ set(runs[0])
    for n in names:
        mean, std, amax, orth = (np.mean([r[n][i] for r in runs]) for i in range(4))
        numel = runs[0][n][4]
        rmean, rstd, rmax, rorth = ref[n]
        if rstd == 0.0:                       # zero / one / constant initialisation
            assert std == 0.0 and mean == rmean, n
            continue
        tol = max(0.03, 4.0 / np.sqrt(2.0 * numel * 4))
        assert abs(std - rstd) <= tol * rstd, (n, std, rstd)
        assert abs(mean - rmean) <= 7.0 * rstd / np.sqrt(numel * 4) + 1e-12, (n, mean, rmean)      # both sides are 4-draw sample means: ~5 sigma of their difference
        if n not in ("_dict.dictionary.weight", "_z_pos.pe"):          # bounded (uniform) initialisers: the range is the bound
            assert abs(amax - rmax) <= max(0.03, 2.0 / numel ** 0.5) * rmax, (n, amax, rmax)
        if rorth >= 0:
            assert orth < 1e-5 and rorth < 1e-5, (n, orth, rorth)


def test_load_before_to_device_stashes_optimizer_state():
    """Base.load() before .to(device) must not raise (the state is applied when the flat buffers exist)"""
    from ocrl_amd import ocrs
    cfg = O.default_cfg(obs_size=16, vocab_size=256, num_slots=3, num_iterations=2, num_dec_blocks=1)
    ocr, env = reference_style_config(cfg)
    m = ocrs.SLATE(ocr, env)
    ck = m.save()
    ck["ocr_opt_state_dict"]["param_groups"][1]["lr"] = 0.125
    m2 = ocrs.SLATE(ocr, env)
    m2.load(ck)
    assert m2._opt.param_groups[1]["lr"] == 0.125 and m2._module._pending_opt is not None
    for (n1, p1), (n2, p2) in zip(m._module.named_parameters(), m2._module.named_parameters()):
        assert n1 == n2 and torch.equal(p1, p2)
