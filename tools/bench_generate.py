"""Latency of the KV-cached autoregressive sampler (ocrl_slate_generate = SLATE_Module._gen_imgs) at the get_samples batch (5 images)."""
import os, sys, time
from types import SimpleNamespace as NS
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import slate_config
from ocrl_amd import ocrs
for S, K in ((64, 6), (128, 6), (256, 16)):
    ocr, env = slate_config(S, num_slots=K)
    torch.manual_seed(0)
    m = ocrs.SLATE(ocr, env); m._module._max_batch = 5; m.to("cuda:0"); m.eval()
    obs = torch.rand(5, 3, S, S, device="cuda")
    m.get_loss(obs, None); m._module._gen_imgs(); torch.cuda.synchronize()
    t0 = time.perf_counter(); m._module._gen_imgs(); torch.cuda.synchronize()
    print(f"_gen_imgs {S}x{S} ({(S // 4) ** 2} tokens, 5 images): {(time.perf_counter() - t0) * 1e3:.1f} ms")
