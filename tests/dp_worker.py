"""One rank of the data-parallel parity test (tests/test_gpu_multi.py).  Started as a fresh process per rank with RANK /
WORLD_SIZE / MASTER_* in the environment (one GPU per rank over RCCL, or with OCRL_DP_ONE_GPU=1 every rank on cuda:0 over gloo); runs one SLATE.update() on its slice of a fixed batch with injected noise and writes the
post-all-reduce gradients, the post-step weights and the reported norm to <out>/rank<r>.pt."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CFG = dict(obs_size=32, vocab_size=512, num_slots=5, num_iterations=2, num_dec_blocks=2)
B_GLOBAL, STEP = 4, 3


def batch():
    from oracle import slate_oracle as O          # the oracle only supplies the closed-form weights / seeded noise here
    cfg = O.default_cfg(**CFG)
    obs = torch.rand(B_GLOBAL, 3, cfg.obs_size, cfg.obs_size, generator=torch.Generator().manual_seed(11))
    return cfg, O.formula_params(cfg), obs, O.make_noise(cfg, B_GLOBAL, 12)


def run_update(cfg, P, obs, noise, device):
    from tests.gpu_util import build_wrapper
    from tests.test_gpu_slate import dev_noise
    model = build_wrapper(cfg, P, device=device)           # eval mode: dropout off, so ranks and the 1-GPU run see the same function
    model._module.inject_noise({k: v.to(device) for k, v in dev_noise(cfg, noise).items()})
    m = model.update(obs.to(device), None, STEP)
    torch.cuda.synchronize()
    eng = model._module.engine
    return dict(g=eng.flat_g.cpu().clone(), p=eng.flat_p.cpu().clone(), norm=float(m["norm"]), loss=float(m["loss"]))


def main():
    out = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    one_gpu = os.environ.get("OCRL_DP_ONE_GPU", "") == "1"      # both ranks on cuda:0, gradients reduced over gloo (RCCL refuses two ranks per device)
    local = 0 if one_gpu else rank
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if one_gpu:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    cfg, P, obs, noise = batch()
    per = B_GLOBAL // world
    sl = slice(rank * per, (rank + 1) * per)
    res = run_update(cfg, P, obs[sl].contiguous(), {k: v[sl].contiguous() for k, v in noise.items()}, dev)
    res["g"] = res["g"] / world          # the buffer holds the SUM; the 1/world mean is folded into the clip kernel
    torch.save(res, os.path.join(out, f"rank{rank}.pt"))
    dist.barrier()
    from ocrl_amd.dist_utils import shutdown
    shutdown()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
