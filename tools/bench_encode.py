"""Latency of the inference path `model(obs)` (SLATE_Module.forward: CNN encoder + slot attention) at RL-style tiny batches."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import slate_config
from ocrl_amd import ocrs

for S in (64, 128):
    ocr, env = slate_config(S)
    torch.manual_seed(0)
    model = ocrs.SLATE(ocr, env)
    model._module._max_batch = 32
    model.to("cuda:0")
    model.eval()
    for B in (1, 8, 32):
        obs = torch.rand(B, 3, S, S, device="cuda")
        for _ in range(5):
            model(obs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            model(obs)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"encode S={S} B={B}: {dt * 1e3:.3f} ms per call, {B / dt:.0f} images/s", flush=True)
