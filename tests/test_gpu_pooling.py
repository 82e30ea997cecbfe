"""GPU parity of the slot-set pooling head (ocrl_pool_transformer_*) against oracle/pooling_oracle.py and the
reference fixtures, through the C ABI and through the drop-in ``poolings`` Python surface."""
import ctypes
import os
import types

import numpy as np
import pytest
import torch

from oracle import pooling_oracle as PO
from tests.gpu_util import log, relerr

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def run_hip(cfg, P, slots, cot, p_drop=0.0, seed=0, want_dslots=True):
    from ocrl_amd import _lib
    L = _lib.lib()
    B, K, Din = slots.shape
    d, h, ff, nl = cfg.d_model, cfg.nhead, cfg.dim_feedforward, cfg.num_layers
    names = [n for n, _ in PO.param_shapes(cfg)]
    w = [P[n].cuda().contiguous() for n in names]
    g = [torch.full_like(t, float("nan")) for t in w]
    pe = PO.pos_table(cfg)
    pos = None if pe is None else pe.cuda().contiguous()
    xs, dc = slots.cuda().contiguous(), cot.cuda().contiguous()
    out = torch.empty(B, d, device="cuda")
    ds = torch.full_like(xs, float("nan")) if want_dslots else None
    n = L.ocrl_pool_transformer_ws_floats(B, K, d, h, ff, nl)
    ws = torch.empty(n, device="cuda")
    arr = (ctypes.c_void_p * len(w))(*[t.data_ptr() for t in w])
    garr = (ctypes.c_void_p * len(g))(*[t.data_ptr() for t in g])
    _lib.check(L.ocrl_pool_transformer_fwd(_lib.ptr(xs), arr, _lib.ptr(pos), _lib.ptr(out), B, K, Din, d, h, ff, nl, p_drop, seed, _lib.ptr(ws), n, None))
    _lib.check(L.ocrl_pool_transformer_bwd(_lib.ptr(xs), _lib.ptr(dc), arr, _lib.ptr(ds), garr, B, K, Din, d, h, ff, nl, p_drop, seed, _lib.ptr(ws), n, None))
    torch.cuda.synchronize()
    return out.cpu(), dict(zip(names, [t.cpu() for t in g])), None if ds is None else ds.cpu()


def hip_masks(cfg, B, p_drop, seed):
    from ocrl_amd import _lib
    L = _lib.lib()
    S, d, ff, h = cfg.num_slots + 1, cfg.d_model, cfg.dim_feedforward, cfg.nhead
    masks = {}
    for l in range(cfg.num_layers):
        for which, (key, shape) in enumerate((("attn", (B, h, S, S)), ("drop1", (B, S, d)), ("ffn", (B, S, ff)), ("drop2", (B, S, d)))):
            m = torch.empty(shape, device="cuda")
            _lib.check(L.ocrl_pool_transformer_dropout_mask(l, which, m.numel(), p_drop, seed, _lib.ptr(m), None))
            masks[f"l{l}.{key}"] = m.cpu()
    return masks


def compare(tag, cfg, got, ref, tol_out=2e-5, tol_g=3e-4):
    out, g, ds = got
    r_out, r_g, r_ds = ref
    e_out = relerr(out, r_out)
    gmax = max(v.abs().max().item() for v in r_g.values())
    rows = sorted(((relerr(g[n], r_g[n], floor=1e-4 * gmax), n) for n in r_g), reverse=True)
    e_ds = relerr(ds, r_ds) if ds is not None else 0.0
    log(f"[pool {tag}] out {e_out:.2e} dslots {e_ds:.2e} grads worst {rows[0][0]:.2e} ({rows[0][1]})")
    assert e_out < tol_out, e_out
    assert e_ds < tol_g, e_ds
    assert rows[0][0] < tol_g, rows[:4]


CASES = [("default", dict(), 3), ("ape_l2", dict(rep_dim=64, num_slots=4, num_layers=2, pos_emb="ape", nhead=4), 2),
         ("k16_b32", dict(num_slots=16, dim_feedforward=512), 32), ("k1", dict(num_slots=1, rep_dim=64, dim_feedforward=256), 5),
         ("d256_h4", dict(d_model=256, nhead=4, dim_feedforward=512, num_layers=3), 4), ("h16", dict(nhead=16, dim_feedforward=256), 2)]


@pytest.mark.parametrize("tag,over,B", CASES)
def test_eval_matches_oracle(tag, over, B):
    cfg = PO.default_cfg(**over)
    P = PO.formula_params(cfg)
    g = torch.Generator().manual_seed(11)
    slots = torch.randn(B, cfg.num_slots, cfg.rep_dim, generator=g)
    cot = torch.randn(B, cfg.d_model, generator=g)
    compare(tag, cfg, run_hip(cfg, P, slots, cot), PO.loss_and_grads(P, slots, cfg, cot))


@pytest.mark.parametrize("tag", ["default", "ape_l2"])
def test_eval_matches_reference_fixture(tag):
    """straight against the numbers the reference module produced (tests/golden/make_golden_pooling.py)"""
    fx = np.load(os.path.join(GOLD, f"pooling_{tag}.npz"))
    rep, K, d, nhead, L, ff, has_pos, B = [int(v) for v in fx["cfg"]]
    cfg = PO.default_cfg(rep_dim=rep, num_slots=K, d_model=d, nhead=nhead, num_layers=L, dim_feedforward=ff, pos_emb="ape" if has_pos else "None")
    out, g, ds = run_hip(cfg, PO.formula_params(cfg), torch.from_numpy(fx["slots"]), torch.from_numpy(fx["cot"]))
    assert relerr(out, torch.from_numpy(fx["out"])) < 2e-5
    assert relerr(ds, torch.from_numpy(fx["dslots"])) < 3e-4
    gmax = max(np.abs(fx["g:" + n][3:]).max() for n in g)
    for n, t in g.items():
        t = t.double().flatten()
        sample = t[:: max(1, t.numel() // 509)][:509].numpy()
        assert np.abs(sample - fx["g:" + n][3:]).max() < 3e-4 * max(np.abs(fx["g:" + n][3:]).max(), 1e-3 * gmax), n


@pytest.mark.parametrize("tag,over,B", [CASES[0], CASES[1], CASES[2]])
def test_train_mode_dropout_parity(tag, over, B):
    """train mode: the oracle consumes the exact keep-masks the kernels derive from (seed, layer, site)"""
    cfg = PO.default_cfg(**over)
    P = PO.formula_params(cfg)
    g = torch.Generator().manual_seed(5)
    slots = torch.randn(B, cfg.num_slots, cfg.rep_dim, generator=g)
    cot = torch.randn(B, cfg.d_model, generator=g)
    p, seed = cfg.dropout, 0x1234567800000009
    masks = hip_masks(cfg, B, p, seed)
    keep = np.mean([m.mean().item() for m in masks.values()])
    assert abs(keep - (1 - p)) < 0.02, keep
    compare(tag + "/train", cfg, run_hip(cfg, P, slots, cot, p, seed), PO.loss_and_grads(P, slots, cfg, cot, masks, p))


def test_detached_slots_and_bad_arguments():
    from ocrl_amd import _lib
    cfg = PO.default_cfg()
    P = PO.formula_params(cfg)
    slots, cot = torch.randn(2, 6, 192), torch.randn(2, 128)
    out, g, ds = run_hip(cfg, P, slots, cot, want_dslots=False)
    assert ds is None and all(torch.isfinite(t).all() for t in g.values())
    L = _lib.lib()
    assert L.ocrl_pool_transformer_fwd(None, None, None, None, 2, 6, 192, 128, 8, 2048, 1, 0.0, 0, None, 0, None) != 0
    x = torch.zeros(8, device="cuda")
    arr = (ctypes.c_void_p * 15)(*[x.data_ptr()] * 15)
    assert L.ocrl_pool_transformer_fwd(_lib.ptr(x), arr, None, _lib.ptr(x), 2, 40, 192, 128, 8, 2048, 1, 0.0, 0, _lib.ptr(x), 8, None) != 0
    assert b"num_slots" in L.ocrl_last_error()
    assert L.ocrl_pool_transformer_fwd(_lib.ptr(x), arr, None, _lib.ptr(x), 2, 6, 192, 128, 8, 2048, 1, 0.0, 0, _lib.ptr(x), 8, None) != 0
    assert b"workspace" in L.ocrl_last_error()


def _pool_cfg(**over):
    c = types.SimpleNamespace(name="Transformer", rep_dim=128, d_model=128, nhead=8, num_layers=1, pos_emb="None", norm_first=False, use_mlp1=False,
                              use_mlp2=False, cw_embedding=False, push_embedding=False, learn_aux_loss=False, learn_downstream_loss=False,
                              ocr_checkpoint=types.SimpleNamespace(run_id="", local_file=""), learning=types.SimpleNamespace(lr=1e-3))
    for k, v in over.items():
        setattr(c, k, v)
    return c


def test_python_surface_trains_with_a_torch_optimizer():
    """Transformer_Module: reference state_dict keys, autograd through the HIP kernels, torch.optim.Adam on its parameters"""
    from ocrl_amd.poolings import Transformer_Module
    cfg = PO.default_cfg()
    m = Transformer_Module(cfg.rep_dim, cfg.num_slots, _pool_cfg()).cuda()
    assert list(m.state_dict()) == [n for n, _ in PO.param_shapes(cfg)]
    P = PO.formula_params(cfg)
    m.load_state_dict(P)
    g = torch.Generator().manual_seed(3)
    slots = torch.randn(4, cfg.num_slots, cfg.rep_dim, generator=g)
    target = torch.randn(4, cfg.d_model, generator=g)
    m.eval()
    out = m(slots.cuda())
    assert relerr(out, PO.forward(P, slots, cfg)) < 2e-5
    with pytest.raises(RuntimeError):
        m(slots)                                                    # CPU tensor: no fallback
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    losses = []
    for _ in range(8):
        opt.zero_grad()
        loss = ((m(slots.cuda()) - target.cuda()) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < 0.8 * losses[0], losses
    a, b = m(slots.cuda()), m(slots.cuda())
    assert (a - b).abs().max().item() > 0                           # train mode: a fresh dropout pattern per call
    m.eval()
    assert (m(slots.cuda()) - m(slots.cuda())).abs().max().item() == 0


def test_pooling_wrapper_and_extractor_over_the_encoder():
    """poolings.Transformer(ocr, config) and OCRExtractor: observations -> SLATE encoder slots -> pooling vector, all on the HIP path"""
    from ocrl_amd import ocrs, poolings
    from ocrl_amd.sb3s import OCRExtractor
    from ocrl_amd.utils.config import compose
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ocr_cfg = compose(os.path.join(root, "configs"), "train_ocr", ["ocr=slate", "ocr.slotattr.num_slots=5", "ocr.dvae.vocab_size=256",
                                                                     "ocr.tfdec.num_dec_blocks=1", "dataset=random-N5C4S4S2", "dataset.obs_size=32"])
    ocr = ocrs.SLATE(ocr_cfg.ocr, ocr_cfg.dataset)
    ocr.to("cuda:0")
    pool = poolings.Transformer(ocr, _pool_cfg())
    pool.to("cuda:0")
    pool.eval()
    obs = torch.rand(4, 3, 32, 32, device="cuda")
    v = pool(obs)
    assert v.shape == (4, 128) and torch.isfinite(v).all()
    slots = ocr(obs)                                                # fresh slot noise per call: pool the same slots on both sides
    P = {k: t.detach().cpu() for k, t in pool._module.state_dict().items()}
    cfg = PO.default_cfg(num_slots=5, rep_dim=ocr.rep_dim)
    assert relerr(pool._module(slots), PO.forward(P, slots.cpu(), cfg)) < 2e-5
    ck = pool.save()
    assert "pooling_module_state_dict" in ck and "ocr_module_state_dict" in ck
    full = types.SimpleNamespace(ocr=ocr_cfg.ocr, env=ocr_cfg.dataset, pooling=_pool_cfg(), num_envs=4, device="cuda:0")
    ex = OCRExtractor(None, full).to("cuda:0")
    ex.eval()
    f = ex(obs)
    assert f.shape == (4, ex.features_dim) and torch.isfinite(f).all()


@pytest.mark.parametrize("flag,widths", [("use_mlp1", [192, 64, 128]), ("use_mlp2", [192, 64, 64, 128])])
def test_slot_mlp_switches(flag, widths):
    """pooling.use_mlp1 / use_mlp2 (poolings/transformer/transformer_module.py:47-63,103-104): the slot MLP in front of the transformer runs
    on the library's GEMM; forward and every MLP gradient against the same chain in fp64 torch around the (separately verified) transformer"""
    from ocrl_amd.poolings import Transformer_Module
    B, K = 5, 6
    base = dict(d_model=128, nhead=8, num_layers=1, pos_emb="None", norm_first=False, use_mlp1=False, use_mlp2=False, cw_embedding=False, push_embedding=False)
    torch.manual_seed(3)
    m = Transformer_Module(192, K, types.SimpleNamespace(**{**base, flag: True})).cuda().eval()
    plain = Transformer_Module(128, K, types.SimpleNamespace(**base)).cuda().eval()
    plain._trans.load_state_dict(m._trans.state_dict())
    slots = torch.randn(B, K, 192, generator=torch.Generator().manual_seed(4))
    # fp64 chain of the reference nn.Sequential with the module's weights
    ws = [(m.mlp[2 * i].weight.detach().double().cpu().requires_grad_(True), m.mlp[2 * i].bias.detach().double().cpu().requires_grad_(True)) for i in range(len(widths) - 1)]
    x = slots.double()
    for i, (w, b) in enumerate(ws):
        x = x @ w.T + b
        if i < len(ws) - 1:
            x = torch.relu(x)
    y_in = x.detach().float().cuda().requires_grad_(True)
    out_ref = plain(y_in)
    out = m(slots.cuda())
    e_out = relerr(out, out_ref)
    dout = torch.randn(B, 128, generator=torch.Generator().manual_seed(5)).cuda()
    out.backward(dout)
    out_ref.backward(dout)
    x.backward(y_in.grad.double().cpu())
    errs = [e_out]
    for i, (w, b) in enumerate(ws):
        errs += [relerr(m.mlp[2 * i].weight.grad, w.grad), relerr(m.mlp[2 * i].bias.grad, b.grad)]
    log(f"[pooling {flag}] output {e_out:.2e}; MLP weight / bias gradients worst {max(errs[1:]):.2e}")
    assert max(errs) < 3e-4, errs


@pytest.mark.parametrize("mode", ["cw", "push"])
def test_gt_state_embeddings_match_reference_fixture(mode):
    """pooling.cw_embedding / push_embedding (transformer_module.py:65-111): the module on the GPU (Linears on the library's GEMM, transformer
    on ocrl_pool_transformer_*) against the reference module's output and embedding-parameter gradients"""
    from ocrl_amd.poolings import Transformer_Module
    fx = np.load(os.path.join(GOLD, f"pooling_{mode}.npz"))
    cfg = PO.default_cfg(rep_dim=128, num_slots=5, d_model=128)
    rc = types.SimpleNamespace(d_model=128, nhead=cfg.nhead, num_layers=cfg.num_layers, pos_emb="None", norm_first=False, use_mlp1=False, use_mlp2=False,
                               cw_embedding=mode == "cw", push_embedding=mode == "push")
    m = Transformer_Module(128, cfg.num_slots, rc)
    G = PO.gt_formula_params(mode)
    sd = m.state_dict()
    assert sorted(k for k in sd if not k.startswith("_trans.")) == sorted(G)          # the reference's parameter names (its `se` buffer is not kept)
    m.load_state_dict({**sd, **PO.formula_params(cfg), **G})
    m = m.cuda().eval()
    out = m(torch.from_numpy(fx["state"]).cuda())
    (out * torch.from_numpy(fx["cot"]).cuda()).sum().backward()
    e_out = relerr(out, torch.from_numpy(fx["out"]))
    named = dict(m.named_parameters())
    gmax = max(np.abs(fx["g:" + n][3:]).max() for n in G)
    worst = 0.0
    for n in G:
        ref = fx["g:" + n]
        t = named[n].grad.double().flatten().cpu()
        got = t[:: max(1, t.numel() // 509)][:509].numpy()
        worst = max(worst, np.abs(got - ref[3:]).max() / max(np.abs(ref[3:]).max(), 1e-3 * gmax))
    log(f"[pooling {mode}_embedding] output {e_out:.2e}; embedding-parameter gradients worst {worst:.2e}")
    assert e_out < 2e-5 and worst < 3e-4


def test_learn_downstream_loss_reaches_the_encoder():
    """pooling.learn_downstream_loss=True (poolings/base.py:53-55): the slots enter the pooling head undetached, so loss.backward() must
    leave d loss / d theta on the encoder (CNN, positional embedding, slot attention), set_zero_grad() / do_step() (poolings/base.py:36-44,
    utils/property_predictor.py:204-211) must step exactly those tensors.  Checked against autograd through the two oracles chained."""
    from oracle import slate_oracle as O
    from ocrl_amd import ocrs, poolings
    from ocrl_amd.utils.config import compose
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ocr_cfg = compose(os.path.join(root, "configs"), "train_ocr", ["ocr=slate", "ocr.slotattr.num_slots=5", "ocr.dvae.vocab_size=256",
                                                                     "ocr.tfdec.num_dec_blocks=1", "dataset=random-N5C4S4S2", "dataset.obs_size=32"])
    torch.manual_seed(11)
    ocr = ocrs.SLATE(ocr_cfg.ocr, ocr_cfg.dataset)
    ocr.to("cuda:0")
    pool = poolings.Transformer(ocr, _pool_cfg(learn_downstream_loss=True))
    pool.to("cuda:0")
    pool.eval()                       # no dropout in the pooling head; the encoder has none on this path
    B, K, D = 3, 5, ocr.rep_dim
    g = torch.Generator().manual_seed(12)
    obs = torch.rand(B, 3, 32, 32, generator=g)
    slot_noise = torch.randn(B, K, D, generator=g)
    cot = torch.randn(B, 128, generator=g)
    before = {k: t.detach().cpu().clone() for k, t in ocr._module.state_dict().items()}
    # ---- HIP: forward through encoder + pooling, backward through both
    pool.set_zero_grad()
    ocr._module.inject_noise(dict(slots=slot_noise.cuda()))
    v = pool(obs.cuda())
    assert v.requires_grad
    (v * cot.cuda()).sum().backward()
    torch.cuda.synchronize()
    eng = ocr._module.engine
    got = {p.name: eng.view(eng.flat_g, p).cpu().clone() for p in eng.params}
    # ---- oracle: the same chain on the CPU with autograd
    cfg = O.default_cfg(obs_size=32, num_slots=5, vocab_size=256, num_dec_blocks=1)
    P = {k: t.clone().requires_grad_(t.dtype.is_floating_point) for k, t in before.items()}
    Pp = {k: t.detach().cpu().clone().requires_grad_(True) for k, t in pool._module.state_dict().items()}
    slots_ref, _ = O.slot_encoder(P, O.cnn_encode(P, obs), slot_noise, cfg)
    v_ref = PO.forward(Pp, slots_ref, PO.default_cfg(num_slots=5, rep_dim=D))
    assert relerr(v, v_ref) < 2e-5
    (v_ref * cot).sum().backward()
    enc = [n for n in got if n.startswith(("_enc.", "_enc_pos.", "_slotattn."))]
    gmax = max(float(P[n].grad.abs().max()) for n in enc)
    errs = {n: relerr(got[n], P[n].grad, floor=1e-4 * gmax) for n in enc}
    worst = max(errs, key=errs.get)
    rest = [n for n in got if n not in enc]
    log(f"[learn_downstream_loss] pooled vector {relerr(v, v_ref):.2e}; encoder gradients through the pooling head: worst {worst}={errs[worst]:.2e} over {len(enc)} tensors; "
        f"{len(rest)} other tensors zero")
    assert errs[worst] < 2e-4, errs
    for n in rest:                     # dVAE, slot projection, transformer decoder: no gradient on this path
        assert P[n].grad is None and not got[n].any(), n
    # pooling-side gradients arrive through torch autograd as before
    for n, p_ in pool._module.named_parameters():
        assert relerr(p_.grad, Pp[n].grad, floor=1e-4 * float(Pp[n].grad.abs().max() + 1e-12)) < 3e-4, n
    # ---- do_step(): Adam on the tensors that have a gradient, nothing else moves (torch's Adam skips parameters whose .grad is None)
    pool.do_step()
    torch.cuda.synchronize()
    after = {k: t.detach().cpu() for k, t in ocr._module.state_dict().items()}
    lr = float(ocr_cfg.ocr.learning.lr_enc)
    for n in enc:
        gr = P[n].grad
        want = before[n] - lr * gr / (gr.abs() + 1e-8)           # first Adam step: m_hat = g, v_hat = g^2
        big = gr.abs() > 1e-3 * gmax                               # elements whose gradient sign is not rounding noise
        if big.any():               # norm_slots.bias: the exact gradient is zero
            assert (after[n] - want)[big].abs().max() <= 1e-3 * lr, n
        assert (after[n] - before[n]).abs().max() <= 1.001 * lr, n      # an Adam step never exceeds lr per element
    for n in rest:
        assert torch.equal(after[n], before[n]), n


def test_extractor_trains_or_freezes_the_encoder_as_get_ocr_does(tmp_path):
    """utils/tools.py:323-347 (get_ocr) decides what sb3s/ocr_extractor.py:33-45 owns: the encoder *module* (no checkpoint, or
    ocr_checkpoint.finetuning) -- its parameters belong to the policy and the RL loss trains them through the slots -- or the wrapper
    object of a pre-trained encoder, which no optimiser sees.  Trainable: a torch optimiser over extractor.parameters() (with its default
    zero_grad(set_to_none=True)) must find gradients on the encoder tensors and only there.  Frozen: the encoder's derived weight images
    are built once (ocrl_slate_freeze_weights) and rebuilt when new weights are loaded."""
    from ocrl_amd import ocrs
    from ocrl_amd.sb3s import OCRExtractor
    from ocrl_amd.utils.config import compose
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ocr_cfg = compose(os.path.join(root, "configs"), "train_ocr", ["ocr=slate", "ocr.slotattr.num_slots=5", "ocr.dvae.vocab_size=256",
                                                                     "ocr.tfdec.num_dec_blocks=1", "dataset=random-N5C4S4S2", "dataset.obs_size=32"])
    obs = torch.rand(4, 3, 32, 32, device="cuda")
    noise = torch.randn(4, 5, 192, device="cuda")
    # ---- trainable: no checkpoint
    full = types.SimpleNamespace(ocr=ocr_cfg.ocr, env=ocr_cfg.dataset, pooling=_pool_cfg(), num_envs=4, device="cuda:0")
    torch.manual_seed(21)
    ex = OCRExtractor(None, full).to("cuda:0")
    ex.train()
    names = [n for n, _ in ex.named_parameters()]
    assert any(n.startswith("_ocr._enc.") for n in names) and any(n.startswith("_pooling.") for n in names)
    opt = torch.optim.Adam([p for p in ex.parameters() if p.dtype.is_floating_point], lr=1e-3)
    with torch.no_grad():
        ex(obs)                                        # sizes the engine for this batch (a new engine hands every parameter a fresh .grad view)
    eng = ex._ocr.engine
    for it in range(2):
        opt.zero_grad()                                # set_to_none=True: every .grad view of the flat buffer is dropped
        before = {n: p.detach().clone() for n, p in ex.named_parameters()}
        ex._ocr.inject_noise(dict(slots=noise))
        f = ex(obs)
        assert f.requires_grad
        f.square().sum().backward()
        with_grad = {n for n, p in ex.named_parameters() if p.grad is not None}
        enc = {n for n in names if n.startswith(("_ocr._enc.", "_ocr._enc_pos.", "_ocr._slotattn."))}
        assert with_grad == enc | {n for n in names if n.startswith("_pooling.")}, sorted(with_grad ^ (enc | {n for n in names if n.startswith("_pooling.")}))[:5]
        for p in eng.params:                           # the published gradients are the flat buffer's
            t = dict(ex._ocr.named_parameters())[p.name]
            if t.grad is not None:
                assert torch.equal(t.grad, eng.view(eng.flat_g, p)) and torch.isfinite(t.grad).all()
        opt.step()
        moved = {n for n, p in ex.named_parameters() if not torch.equal(p.detach(), before[n])}
        assert "_ocr._enc._encoder.0.m.weight" in moved and "_ocr._slotattn.slot_attention.gru.weight_hh" in moved
        assert not any(n.startswith(("_ocr._dvae.", "_ocr._tfdec.", "_ocr._slotproj.", "_ocr._dict.")) for n in moved)
    # the optimiser wrote through the parameter views into the flat buffer the kernels read
    assert torch.equal(dict(ex._ocr.named_parameters())["_enc._encoder.0.m.weight"].detach().flatten(),
                       eng.view(eng.flat_p, next(p for p in eng.params if p.name == "_enc._encoder.0.m.weight")).flatten())
    # ---- frozen: pre-trained checkpoint, no finetuning
    src = ocrs.SLATE(ocr_cfg.ocr, ocr_cfg.dataset); src.to("cuda:0")
    path = str(tmp_path / "model_latest.pth")
    torch.save(src.save(), path)
    pc = _pool_cfg(ocr_checkpoint=types.SimpleNamespace(run_id="", local_file=path, finetuning=False))
    fx = OCRExtractor(None, types.SimpleNamespace(ocr=ocr_cfg.ocr, env=ocr_cfg.dataset, pooling=pc, num_envs=4, device="cuda:0")).to("cuda:0")
    fx.eval()
    assert not any(n.startswith("_ocr.") for n, _ in fx.named_parameters())          # the wrapper is not a module: nothing to train
    outs = []
    for _ in range(3):                                 # call 1 builds the weight images, calls 2 and 3 re-use them
        fx._ocr._module.inject_noise(dict(slots=noise))
        outs.append(fx(obs))
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    src._module.inject_noise(dict(slots=noise))
    want = fx._pooling(src(obs))                       # the unfrozen wrapper with the same weights
    assert torch.equal(outs[0], want)
    # new weights loaded while frozen: picked up at the next call
    sd = {k: (v * 1.5 if k == "_enc._encoder.3.weight" else v) for k, v in src._module.state_dict().items()}
    fx._ocr._module.load_state_dict(sd)
    fx._ocr._module.inject_noise(dict(slots=noise))
    assert not torch.equal(fx(obs), outs[0])
    # finetuning a pre-trained encoder: the module again
    pc2 = _pool_cfg(ocr_checkpoint=types.SimpleNamespace(run_id="", local_file=path, finetuning=True))
    tx = OCRExtractor(None, types.SimpleNamespace(ocr=ocr_cfg.ocr, env=ocr_cfg.dataset, pooling=pc2, num_envs=4, device="cuda:0")).to("cuda:0")
    assert any(n.startswith("_ocr._enc.") for n, _ in tx.named_parameters())
    assert torch.equal(tx._ocr.engine.flat_p, src._module.engine.flat_p)
