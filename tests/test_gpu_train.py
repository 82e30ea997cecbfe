"""train_ocr.py end to end on the GPU at BASELINE configs[0]'s shape (64x64, batch 4): runs, logs, checkpoints,
resumes; the checkpoint keeps the reference's dict layout and round-trips through SLATE.save()/load()."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_ocr_runs_checkpoints_and_resumes(tmp_path):
    import train_ocr
    run = str(tmp_path / "run")
    args = ["ocr=slate", "ocr.slotattr.num_slots=6", "ocr.slotattr.num_iterations=3", "dataset=random-N5C4S4S2", "device=cuda:0",
            "batch_size=4", "num_workers=0", "dataset.synthetic_train=64", "dataset.synthetic_val=8", "eval_interval=5",
            "max_steps=10", "log_interval=1", f"run_dir={run}"]
    assert train_ocr.main(args) == 10
    lines = [json.loads(l) for l in open(os.path.join(run, "metrics.jsonl"))]
    tr = [l for l in lines if "train/loss" in l]
    assert len(tr) == 10 and all(torch.isfinite(torch.tensor(l["train/loss"])) for l in tr)
    assert {"train/dvae_mse", "train/cross_entropy", "train/tau", "train/lr_dvae", "train/lr_enc", "train/lr_dec", "train/norm"} <= set(tr[0])
    assert any("val/loss" in l for l in lines)
    ck = torch.load(os.path.join(run, "checkpoints", "model_latest.pth"), weights_only=True)
    assert set(ck) == {"step", "epoch", "best_val_loss", "ocr_module_state_dict", "ocr_opt_state_dict"} and ck["step"] == 10
    assert os.path.exists(os.path.join(run, "checkpoints", "model_5.pth")) and os.path.exists(os.path.join(run, "checkpoints", "model_best.pth"))
    st = ck["ocr_opt_state_dict"]["state"]
    assert len(st) > 100 and float(next(iter(st.values()))["step"]) == 10.0
    # resume continues from step 10
    args[args.index("max_steps=10")] = "max_steps=12"
    assert train_ocr.main(args) == 12
    assert tr[0]["train/loss"] > tr[-1]["train/loss"] * 0.5          # loss is not exploding


def test_checkpoint_round_trip_restores_weights_and_adam_state():
    from ocrl_amd import ocrs
    from ocrl_amd.utils.config import compose
    from train_ocr import ROOT
    c = compose(os.path.join(ROOT, "configs"), "train_ocr", ["ocr=slate", "ocr.dvae.vocab_size=256", "ocr.tfdec.num_dec_blocks=1",
                                                              "dataset=random-N5C4S4S2", "dataset.obs_size=16"])
    obs = torch.rand(2, 3, 16, 16, device="cuda")
    a = ocrs.SLATE(c.ocr, c.dataset); a.to("cuda:0"); a.train()
    for s in range(3):
        a.update(obs, None, s)
    ck = a.save()
    b = ocrs.SLATE(c.ocr, c.dataset); b.to("cuda:0"); b.train()
    b.load(ck)
    for (n1, p1), (n2, p2) in zip(a._module.named_parameters(), b._module.named_parameters()):
        assert n1 == n2 and torch.equal(p1, p2), n1
    a._module.set_seed(7); b._module.set_seed(7)
    ma, mb = a.update(obs, None, 3), b.update(obs, None, 3)
    assert float(ma["loss"]) == float(mb["loss"])
    # every reduction on the path runs in a fixed order (no float atomics): a restored run continues bit for bit
    for (n1, p1), (n2, p2) in zip(a._module.named_parameters(), b._module.named_parameters()):
        assert torch.equal(p1, p2), n1


def test_train_ocr_runs_iodine(tmp_path):
    """BASELINE config 4 plumbing: `train_ocr.py ocr=iodine` with masks from the synthetic dataset (ARI is reported)"""
    import train_ocr
    run = str(tmp_path / "run_iodine")
    args = ["ocr=iodine", "ocr.num_slots=7", "ocr.num_iterations=5", "dataset=random-N5C4S4S2", "dataset.with_masks=True", "dataset.obs_size=32",
            "device=cuda:0", "batch_size=4", "num_workers=0", "dataset.synthetic_train=32", "dataset.synthetic_val=8", "eval_interval=4",
            "max_steps=8", "log_interval=1", f"run_dir={run}"]
    assert train_ocr.main(args) == 8
    lines = [json.loads(l) for l in open(os.path.join(run, "metrics.jsonl"))]
    tr = [l for l in lines if "train/loss" in l]
    assert len(tr) == 8 and all(torch.isfinite(torch.tensor(l["train/loss"])) for l in tr)
    assert {"train/mse", "train/ari", "train/kld", "train/norm"} <= set(tr[0])
    assert tr[-1]["train/loss"] < tr[0]["train/loss"]
    ck = torch.load(os.path.join(run, "checkpoints", "model_latest.pth"), weights_only=True)
    assert "refine.lstm.weight_ih" in ck["ocr_module_state_dict"] and ck["step"] == 8


def test_cabi_rccl_allreduce_single_rank():
    """ocrl_comm_* (lazy librccl): world = 1 all-reduce leaves the buffer unchanged; N > 1 needs one GPU per rank (driver's scaling run)"""
    import ctypes
    from ocrl_amd import _lib
    L = _lib.lib()
    uid = ctypes.create_string_buffer(128)
    torch.cuda.set_device(0)
    _lib.check(L.ocrl_comm_unique_id(uid, 128))
    h = ctypes.c_void_p()
    _lib.check(L.ocrl_comm_init(ctypes.byref(h), 0, 1, uid))
    assert L.ocrl_comm_world(h) == 1
    x = torch.arange(1024, dtype=torch.float32, device="cuda")
    ref = x.clone()
    _lib.check(L.ocrl_comm_allreduce(h, _lib.ptr(x), x.numel(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert torch.equal(x, ref)
    L.ocrl_comm_destroy(h)
