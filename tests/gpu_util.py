"""Helpers shared by the GPU parity tests."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")


def log(msg):
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "parity_log.txt"), "a") as f:
        f.write(msg + "\n")
    print(msg)
    sys.stdout.flush()


def relerr(a, b, floor=0.0):
    """max |a-b| / max(max|b|, floor)"""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    den = max(b.abs().max().item(), floor, 1e-30)
    return ((a - b).abs().max().item()) / den


def dims_from_cfg(cfg):
    from types import SimpleNamespace
    return SimpleNamespace(obs_size=cfg.obs_size, obs_channels=cfg.obs_channels, vocab_size=cfg.vocab_size, d_model=cfg.d_model,
                           cnn_hidden=cfg.cnn_hidden, num_slots=cfg.num_slots, num_iterations=cfg.num_iterations,
                           slot_size=cfg.slot_size, mlp_hidden=cfg.mlp_hidden, num_dec_blocks=cfg.num_dec_blocks,
                           num_dec_heads=cfg.num_dec_heads, dropout=cfg.dropout, use_bcdec=cfg.use_bcdec, hard=cfg.hard)


def load_params(engine, P):
    """oracle param dict (reference state_dict names) -> engine flat buffer"""
    for p in engine.params:
        engine.view(engine.flat_p, p).copy_(P[p.name].to(engine.device).reshape(p.shape))
    torch.cuda.synchronize()
