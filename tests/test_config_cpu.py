"""Host logic on CPU: the Hydra-grammar resolver, the synthetic dataset, ARI, schedules, the API surface."""
import os

import numpy as np
import pytest
import torch

from ocrl_amd.utils.config import compose

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "configs")


def test_compose_reference_command_line():
    c = compose(CFG, "train_ocr", ["ocr=slate", "ocr.slotattr.num_slots=6", "ocr.slotattr.num_iterations=3",
                                   "dataset=random-N5C4S4S2", "device=cuda:0", "tags=slate", "dataset.obs_size=128", "batch_size=4"])
    assert c.ocr.name == "SLATE" and c.ocr.slotattr.num_slots == 6 and c.ocr.dvae.vocab_size == 4096
    assert c.ocr.learning.clip == 0.05 and c.ocr.learning.lr_dvae == pytest.approx(3e-4)
    assert c.dataset.obs_size == 128 and c.dataset.obs_channels == 3 and c.dataset.name == "RandomN5C4S4S2"
    assert c.batch_size == 4 and c.eval_interval == 1000 and c.max_epochs == 1000 and c.seed == 0
    assert c.run_dir.endswith("SLATE-RandomN5C4S4S2")
    # DictConfig-style probes used by the reference (ocrs/base.py:21-22,65-66)
    assert hasattr(c.ocr, "learning") and not hasattr(c.ocr.learning, "lr") and hasattr(c.ocr.learning, "clip")
    assert c.dataset.dataset_dir == "datasets"          # inherited through _synthetic_env_base -> _base


def test_mandatory_groups_and_bad_overrides():
    with pytest.raises(ValueError, match="You must specify 'ocr'"):
        compose(CFG, "train_ocr", ["dataset=random-N5C4S4S2"])
    with pytest.raises(KeyError):
        compose(CFG, "train_ocr", ["ocr=slate", "dataset=random-N5C4S4S2", "ocr.nonexistent=1"])
    c = compose(CFG, "train_ocr", ["ocr=slate", "dataset=random-N5C4S4S2", "+ocr.extra=7"])
    assert c.ocr.extra == 7
    with pytest.raises(FileNotFoundError):
        compose(CFG, "train_ocr", ["ocr=nope", "dataset=random-N5C4S4S2"])


def test_slotattn_alias_config():
    c = compose(CFG, "train_ocr", ["ocr=slotattn", "dataset=random-N5C4S4S2"])
    assert c.ocr.name == "SLATE" and c.ocr.use_bcdec is True and c.ocr.slotattr.slot_size == 192


def test_synthetic_scenes_follow_the_env_spec():
    from ocrl_amd.utils.data import random_sprite_scenes
    img, masks = random_sprite_scenes(6, 64, seed=3, with_masks=True)
    assert img.shape == (6, 64, 64, 3) and img.dtype == np.uint8
    assert masks.shape == (6, 6, 64, 64, 1)
    assert np.allclose(masks.sum(1), 1.0)                              # a partition: objects + background
    assert np.array_equal(img, random_sprite_scenes(6, 64, seed=3))    # deterministic
    colours = {tuple(c) for c in img.reshape(-1, 3)}
    assert colours <= {(0, 0, 0), (0, 0, 255), (0, 255, 0), (255, 255, 0), (255, 0, 0)}
    assert (masks[:, -1].mean() > 0.5)                                 # mostly background


def test_dataset_and_loader_sample_format():
    from ocrl_amd.utils.datasets import get_dataloaders
    c = compose(CFG, "train_ocr", ["ocr=slate", "dataset=random-N5C4S4S2", "dataset.synthetic_train=16", "dataset.synthetic_val=8",
                                   "dataset.with_masks=True"])
    tr, va = get_dataloaders(c.dataset, 4, 0, raw_uint8=False)          # the reference's sample format (utils/datasets.py:13-24)
    b = next(iter(tr))
    assert b["obss"].shape == (4, 3, 64, 64) and b["obss"].dtype == torch.float32 and 0 <= b["obss"].min() and b["obss"].max() <= 1
    assert b["masks"].permute(0, 1, 4, 2, 3).shape == (4, 6, 1, 64, 64)
    assert len(va.dataset) == 8
    # default: the stored uint8 HWC image travels, the conversion runs on the GPU; both describe the same pixels
    tr8, _ = get_dataloaders(c.dataset, 4, 0)
    b8 = next(iter(tr8))
    assert b8["obss_u8"].shape == (4, 64, 64, 3) and b8["obss_u8"].dtype == torch.uint8 and "obss" not in b8
    s0, s8 = tr.dataset[3], tr8.dataset[3]
    assert torch.equal(s8["obss_u8"].permute(2, 0, 1).float() / 255.0, s0["obss"]) and torch.equal(s0["masks"], s8["masks"])


def test_ari_matches_sklearn():
    from sklearn.metrics import adjusted_rand_score
    from ocrl_amd.utils.tools import _adjusted_rand_score
    rs = np.random.RandomState(0)
    for _ in range(5):
        a, b = rs.randint(0, 5, 500), rs.randint(0, 4, 500)
        assert _adjusted_rand_score(a, b) == pytest.approx(adjusted_rand_score(a, b), abs=1e-12)
    assert _adjusted_rand_score([0, 0, 1, 1], [1, 1, 0, 0]) == pytest.approx(1.0)


def test_slate_surface_matches_reference_names():
    from oracle import slate_oracle as O
    from ocrl_amd import ocrs
    c = compose(CFG, "train_ocr", ["ocr=slate", "ocr.slotattr.num_slots=6", "dataset=random-N5C4S4S2"])
    m = ocrs.SLATE(c.ocr, c.dataset)
    assert m.name == "SLATE" and m.rep_dim == 192 and m.num_slots == 6
    sd = m._module.state_dict()
    cfg = O.default_cfg(obs_size=64, num_slots=6)
    assert {k for k in sd if not k.endswith("linear_position_embedding")} == {n for n, _, _, _ in O.param_shapes(cfg)}
    for n, shp, _, _ in O.param_shapes(cfg):
        assert tuple(sd[n].shape) == tuple(shp), n
    names = {id(p): n for n, p in m._module.named_parameters()}
    for g in range(3):
        assert [names[id(p)] for p in m._opt.param_groups[g]["params"]] == [n for n, _, gg, _ in O.param_shapes(cfg) if gg == g]
    ck = m.save()
    assert set(ck) == {"ocr_module_state_dict", "ocr_opt_state_dict"} and set(ck["ocr_opt_state_dict"]) == {"state", "param_groups"}
    with pytest.raises(RuntimeError):
        m.to("cpu")
    sa = ocrs.SLATE(compose(CFG, "train_ocr", ["ocr=slotattn", "dataset=random-N5C4S4S2"]).ocr, c.dataset)
    assert "_dec._decoder.3.weight" in sa._module.state_dict() and "_dec._pos_emb.linear_position_embedding" not in sd


def test_schedules_match_oracle():
    from oracle import slate_oracle as O
    from ocrl_amd.ocrs.slate import cosine_anneal, linear_warmup
    for step in (0, 1, 999, 15000, 29999, 30000, 50000):
        assert cosine_anneal(step, 1.0, 0.1, 0, 30000) == O.cosine_anneal(step, 1.0, 0.1, 0, 30000)
        assert linear_warmup(step, 0, 1, 0, 30000) == O.linear_warmup(step, 0, 1, 0, 30000)


def test_pooling_module_mirrors_reference_state_dict():
    """poolings.Transformer_Module holds its parameters under the reference's names / shapes (checkpoints load unchanged) and refuses a
    CPU tensor instead of falling back (no compute here)."""
    import types
    import torch
    from oracle import pooling_oracle as PO
    from ocrl_amd.poolings import Transformer_Module
    for over, pos in ((dict(), "None"), (dict(num_layers=2, nhead=4, num_slots=4, rep_dim=64), "ape")):
        cfg = PO.default_cfg(pos_emb=pos, **over)
        pc = types.SimpleNamespace(d_model=cfg.d_model, nhead=cfg.nhead, num_layers=cfg.num_layers, pos_emb=pos, norm_first=False, use_mlp1=False,
                                   use_mlp2=False, cw_embedding=False, push_embedding=False)
        m = Transformer_Module(cfg.rep_dim, cfg.num_slots, pc)
        sd = m.state_dict()
        names = [(n, tuple(s)) for n, s in PO.param_shapes(cfg)]
        assert [(k, tuple(v.shape)) for k, v in sd.items() if not k.endswith(".pe")] == names
        if pos == "ape":
            assert tuple(sd["_trans._pos.pe"].shape) == (cfg.num_slots + 1, 1, cfg.d_model)
            assert torch.allclose(sd["_trans._pos.pe"][:, 0], PO.pos_table(cfg))
        with pytest.raises(RuntimeError):
            m(torch.zeros(2, cfg.num_slots, cfg.rep_dim))
    # cw_embedding / push_embedding (transformer_module.py:65-78): the reference's parameter names; its materialised sinusoid table `se` is
    # evaluated on the fly and ignored when a reference checkpoint carries it
    mc = Transformer_Module(64, 4, types.SimpleNamespace(d_model=128, nhead=8, num_layers=1, pos_emb="None", cw_embedding=True))
    assert [(k, tuple(v.shape)) for k, v in mc.state_dict().items()][:4] == [("arm_emb.weight", (128, 3584)), ("arm_emb.bias", (128,)), ("obj_emb.weight", (128, 387)), ("obj_emb.bias", (128,))]
    mc.load_state_dict({**mc.state_dict(), "cw_emb.se": torch.zeros(10001, 1, 128)})
    mp = Transformer_Module(64, 4, types.SimpleNamespace(d_model=128, nhead=8, num_layers=1, pos_emb="None", push_embedding=True))
    assert [k for k in mp.state_dict() if not k.startswith("_trans.")] == ["color_emb.weight", "shape_emb.weight", "obj_emb.weight", "obj_emb.bias"]
    with pytest.raises(ValueError):
        Transformer_Module(64, 4, types.SimpleNamespace(d_model=256, nhead=8, num_layers=1, pos_emb="None", cw_embedding=True))
    # use_mlp1 / use_mlp2 (transformer_module.py:47-63): the slot MLP's keys and shapes are the reference nn.Sequential's
    m1 = Transformer_Module(192, 6, types.SimpleNamespace(d_model=128, nhead=8, num_layers=1, pos_emb="None", use_mlp1=True))
    assert [(k, tuple(v.shape)) for k, v in m1.state_dict().items()][:4] == [("mlp.0.weight", (64, 192)), ("mlp.0.bias", (64,)), ("mlp.2.weight", (128, 64)), ("mlp.2.bias", (128,))]
    assert tuple(m1.state_dict()["_trans._linear.weight"].shape) == (128, 128)
    m2 = Transformer_Module(192, 6, types.SimpleNamespace(d_model=128, nhead=8, num_layers=1, pos_emb="None", use_mlp2=True))
    assert [k for k in m2.state_dict() if k.startswith("mlp.")] == ["mlp.0.weight", "mlp.0.bias", "mlp.2.weight", "mlp.2.bias", "mlp.4.weight", "mlp.4.bias"]


def test_multi_head_slot_attention_shapes():
    """ocr.slotattr.num_slot_heads (ocrs/common/slot_attn.py:28,54-92) reaches the backend; shapes whose heads * slots columns do not fit one
    16-column MFMA tile raise instead of silently running something else"""
    from ocrl_amd import ocrs
    c = compose(CFG, "train_ocr", ["ocr=slate", "dataset=random-N5C4S4S2", "ocr.slotattr.num_slot_heads=2"])
    m = ocrs.SLATE(c.ocr, c.dataset)                 # 6 slots x 2 heads: supported
    assert m._module._dims.num_slot_heads == 2
    for over in (["ocr.slotattr.num_slot_heads=4"],                                   # 6 x 4 = 24 columns
                 ["ocr.slotattr.num_slot_heads=2", "ocr.slotattr.num_slots=10"],      # more than 8 slots
                 ["ocr.slotattr.num_slot_heads=5"]):                                  # 192 / 5
        c = compose(CFG, "train_ocr", ["ocr=slate", "dataset=random-N5C4S4S2"] + over)
        with pytest.raises(NotImplementedError, match="num_slot_heads"):
            ocrs.SLATE(c.ocr, c.dataset)
    c = compose(CFG, "train_ocr", ["ocr=slate", "dataset=random-N5C4S4S2"])
    assert c.ocr.slotattr.num_slot_heads == 1
    ocrs.SLATE(c.ocr, c.dataset)           # the shipped configuration constructs (CPU container tensors; .to(cuda) is needed to run)
