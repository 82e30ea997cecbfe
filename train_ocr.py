"""Pre-training entry point with the reference's command line (train_ocr.py:18-116 of the reference):

    python train_ocr.py ocr=slate ocr.slotattr.num_slots=6 ocr.slotattr.num_iterations=3 dataset=random-N5C4S4S2 device=cuda:0
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train_ocr.py ocr=slate dataset=random-N5C4S4S2 ...

Same loop semantics (epoch/step loop, eval every eval_interval steps, best tracking, checkpoints model_<step>.pth /
model_latest.pth / model_best.pth with keys step, epoch, best_val_loss, ocr_module_state_dict, ocr_opt_state_dict,
resume from load.resume_checkpoint or the run directory's latest).  The encoder runs on the HIP backend
(ocrl_amd.ocrs); under torchrun each rank drives one GPU and gradients are all-reduced over RCCL.
"""
import json
import logging
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from ocrl_amd import ocrs  # noqa: E402
from ocrl_amd.utils.config import compose  # noqa: E402
from ocrl_amd.utils.datasets import get_dataloaders  # noqa: E402
from ocrl_amd.utils.tools import get_item, obs_from_uint8, to_device  # noqa: E402

log = logging.getLogger("train_ocr")


def load(model, run_dir, resume_checkpoint=None):
    """utils/tools.py:223-263: explicit checkpoint > <run>/checkpoints/model_latest.pth > fresh"""
    ckpt = None
    dev = model._module.engine.device
    if resume_checkpoint is not None:
        ckpt = torch.load(resume_checkpoint, map_location=dev, weights_only=True)
    elif (latest := Path(run_dir) / "checkpoints" / "model_latest.pth").exists():
        ckpt = torch.load(latest, map_location=dev, weights_only=True)
    if ckpt is None:
        return 0, 0, 1e10
    model.load(ckpt)
    return ckpt["step"], ckpt["epoch"], ckpt["best_val_loss"]


def save(model, run_dir, step=0, epoch=0, best_val_loss=1e5, best=False):
    """utils/tools.py:267-289"""
    d = Path(run_dir) / "checkpoints"
    d.mkdir(parents=True, exist_ok=True)
    ckpt = {"step": step, "epoch": epoch, "best_val_loss": best_val_loss}
    ckpt.update(model.save())
    torch.save(ckpt, d / f"model_{step}.pth")
    torch.save(ckpt, d / "model_latest.pth")
    if best:
        torch.save(ckpt, d / "model_best.pth")


def batch_inputs(batch, device):
    """train_ocr.py:52-53 + utils/datasets.py:17: the uint8 images are uploaded as stored and converted (HWC -> CHW, /255) on the GPU"""
    obs = obs_from_uint8(to_device(batch["obss_u8"], device)) if "obss_u8" in batch else to_device(batch["obss"], device)
    masks = to_device(batch["masks"].permute(0, 1, 4, 2, 3), device) if "masks" in batch else None
    return obs, masks


def eval_and_save(model, val_dl, epoch, step, best_val_loss, config, logger, is_main):
    """train_ocr.py:72-116 of the reference"""
    metrics = []
    batch = None
    for batch in val_dl:
        obs, masks = batch_inputs(batch, config.device)
        metrics.append({k: get_item(v) for k, v in model.get_loss(obs, masks).items()})
    out = {k: float(np.mean([np.mean(m[k]) for m in metrics])) for k in metrics[0]}
    best = out["loss"] < best_val_loss
    if best:
        best_val_loss = out["loss"]
    out["best_loss"] = best_val_loss
    logger({f"val/{k}": v for k, v in out.items()}, step)
    log.info(f"[Epoch {epoch}, Step {step}] " + " / ".join(f"val/{k} {v:.4f}" for k, v in out.items()))
    if is_main:
        if best and batch is not None:
            obs, _ = batch_inputs(batch, config.device)
            samples = model.get_samples(obs[: config.num_visualization])
            np.save(Path(config.run_dir) / f"samples_{step}.npy", samples["samples"])
        save(model, config.run_dir, step=step, epoch=epoch, best_val_loss=best_val_loss, best=best)
    return best_val_loss


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    config = compose(os.path.join(ROOT, "configs"), "train_ocr", argv)
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    if world > 1:
        import torch.distributed as dist
        config.device = f"cuda:{local}"
        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(config.device))
    is_main = rank == 0
    logging.basicConfig(level=logging.INFO if is_main else logging.WARNING, format="%(asctime)s %(message)s")
    os.makedirs(config.run_dir, exist_ok=True)
    logf = open(os.path.join(config.run_dir, "metrics.jsonl"), "a") if is_main else None

    def logger(d, step):
        if logf is not None:
            logf.write(json.dumps({"step": step, **{k: (float(v) if np.ndim(v) == 0 else np.asarray(v).tolist()) for k, v in d.items()}}) + "\n")
            logf.flush()

    np.random.seed(config.seed)
    torch.manual_seed(config.seed)      # identical initial weights on every rank
    train_dl, val_dl = get_dataloaders(config.dataset, config.batch_size, config.num_workers, rank, world, config.seed)
    model = getattr(ocrs, config.ocr.name)(config.ocr, config.dataset)
    model._module._max_batch = config.batch_size
    model.to(config.device)
    model._module.set_seed(config.seed * 1000 + rank)
    step, epoch, best_val_loss = load(model, config.run_dir, config.load.resume_checkpoint)
    log.info(f"training {config.ocr.name} on {config.dataset.name}: batch {config.batch_size} x {world} GPU(s), start step {step}")

    t0, n_img = time.perf_counter(), 0
    done = False
    while epoch < config.max_epochs and not done:
        model.train()
        if world > 1 and hasattr(train_dl.sampler, "set_epoch"):
            train_dl.sampler.set_epoch(epoch)
        for batch in train_dl:
            obs, masks = batch_inputs(batch, config.device)
            metrics = model.update(obs, masks, step)
            n_img += obs.shape[0] * world
            if step % config.log_interval == 0:
                vals = {f"train/{k}": get_item(v) for k, v in metrics.items()}
                vals["train/images_per_sec"] = n_img / (time.perf_counter() - t0)
                logger(vals, step)
                log.info(f"step {step} loss {float(np.mean(vals['train/loss'])):.4f} ({vals['train/images_per_sec']:.1f} img/s)")
            step += 1
            if step % config.eval_interval == 0:
                model.eval()
                best_val_loss = eval_and_save(model, val_dl, epoch, step, best_val_loss, config, logger, is_main)
                model.train()
            if config.max_steps is not None and step >= config.max_steps:
                done = True
                break
        epoch += 1
    if is_main and (config.max_steps is None or step % config.eval_interval != 0):
        save(model, config.run_dir, step=step, epoch=epoch, best_val_loss=best_val_loss, best=False)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return step


if __name__ == "__main__":
    main()
