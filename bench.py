"""Headline benchmark: images/sec of one full SLATE pre-training step (forward + backward + gradient
all-reduce + inf-norm clip + Adam; train mode, dropout 0.1, on-device RNG) at 128x128, 6 slots, 3 iterations,
vocab 4096 — BASELINE.json's metric — on N MI355X (one process per GPU, RCCL via torch.distributed).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events on the launch stream around the
dominant kernel family (the 5x5 / 64-channel MFMA convolution, forward and backward-data launches);
`cpu_baseline` times the CPU oracle (oracle/slate_oracle.py, a port of the reference step) on the host cores.
"""
import argparse
import ctypes
import json
import os
import sys
import time
from types import SimpleNamespace as NS

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F32_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: fp32-input MFMA dense peak
FWD_FLOP_PER_IMAGE = {(128, 6): 22384148480, (64, 6): 5004001280, (256, 16): 129805844480}     # (S, slots), 3 iters


def slate_config(obs_size, num_slots=6, num_iterations=3):
    """configs/ocr/slate.yaml with BASELINE's overrides"""
    ocr = NS(name="SLATE", tau_start=1.0, tau_final=0.1, tau_steps=30000, hard=False, use_cnn_feat=False, use_bcdec=False,
             dvae=NS(vocab_size=4096, d_model=192), cnn=NS(hidden_size=64),
             slotattr=NS(num_iterations=num_iterations, num_slots=num_slots, num_slot_heads=1, slot_size=192, mlp_hidden_size=192, pos_channels=4),
             tfdec=NS(num_dec_blocks=4, num_dec_heads=4),
             learning=NS(lr_half_life=250000, lr_dvae=3e-4, lr_enc=1e-4, lr_dec=3e-4, lr_warmup_steps=30000, dropout=0.1, clip=0.05))
    env = NS(obs_size=obs_size, obs_channels=3)
    return ocr, env


def cpu_baseline(obs_size, batch, steps):
    """the CPU oracle's update() (port of ocrs/slate/slate.py:53-69 + ocrs/base.py:60-74) on the host cores"""
    from oracle import slate_oracle as O
    from ocrl_amd.utils.data import random_sprite_scenes, scenes_to_obs
    try:
        cores = len(os.sched_getaffinity(0))      # the cores this process may actually use (not the host's total)
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline: oracle update() on {cores} host threads, batch {batch}, {steps} timed steps", file=sys.stderr, flush=True)
    cfg = O.default_cfg(obs_size=obs_size, num_slots=6, num_iterations=3)
    tr = O.OracleTrainer(cfg, O.formula_params(cfg))
    obs = scenes_to_obs(random_sprite_scenes(batch, obs_size, seed=123))
    times = []
    for step in range(steps + 1):
        t0 = time.perf_counter()
        noise = O.make_noise(cfg, batch, step)           # the reference draws these inside the step (RNG-bound on CPU)
        masks = O.make_masks(cfg, batch, 1000 + step)
        tr.update(obs, noise, step, masks)
        times.append(time.perf_counter() - t0)
        print(f"[bench] cpu_baseline step {step}: {times[-1]:.1f} s", file=sys.stderr, flush=True)
    t = sum(times[1:]) / steps
    return {"value": round(batch / t, 4), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle update() SLATE {obs_size}x{obs_size}/6 slots/3 iters, batch {batch}, train mode dropout 0.1, "
                      f"1 warm-up + {steps} timed steps, {t:.2f} s/step"}


def bench_iodine(args, dev, dist, rank, world):
    """BASELINE config 4: IODINE update() = _forward + backward + all-reduce + L2 clip + Adam (the per-step CPU ARI metric of
    get_loss is excluded from the timed region, like SLATE's masks=None path)."""
    from ocrl_amd import ocrs
    from ocrl_amd.dist_utils import allreduce_grads_
    from ocrl_amd.utils.data import random_sprite_scenes, scenes_to_obs
    S, B, K = args.obs_size, args.batch, args.num_slots
    ocr = NS(name="Iodine", slot_size=64, num_iterations=5, num_slots=K, img_channels=3, sigma=0.35, beta=1.0, layer_norm=True,
             ref_cnn_hidden_size=64, ref_mlp_hidden_size=256, ref_cnn_layers=4, ref_cnn_kernel_size=3, ref_cnn_stride_size=2,
             dec_cnn_hidden_size=64, dec_cnn_layers=4, dec_cnn_kernel_size=3, learning=NS(lr=3e-4, clip=5.0, clip_norm_type=2.0))
    torch.manual_seed(0)
    model = ocrs.Iodine(ocr, NS(obs_size=S, obs_channels=3))
    model._module._max_batch = B
    model.to(dev)
    model.train()
    model._module.set_seed(1 + rank)
    pool = [scenes_to_obs(random_sprite_scenes(B, S, seed=1000 * rank + i)).to(dev) for i in range(4)]
    mod, lr = model._module, ocr.learning

    def step_fn(i):
        out = mod._forward(pool[i % len(pool)])
        mod.backward()
        scale = allreduce_grads_(mod.engine.flat_g)
        model._opt.step(lr.clip, scale)
        return out[4]

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()

    n = 0
    for _ in range(args.warmup):
        step_fn(n); n += 1
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step_fn(n); n += 1
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        ips = B * world * args.steps / dt
        # decoder 1.236 GF per (image, slot, iteration) forward at 64x64: fwd I, in-forward bwd-data I-1, bwd 2I
        dec = 2.0 * 9 * (66 * 64 + 3 * 64 * 64 + 64 * 4) * S * S
        flop_img = dec * K * (5 + 4 + 2 * 5)
        print(json.dumps({
            "metric": f"images/sec (node) IODINE pretrain {S}x{S}, {K} slots, 5 iters", "value": round(ips, 2), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"IODINE {S}x{S}, {K} slots, 5 refinement iterations; _forward + backward + all-reduce + L2 clip + Adam "
                                   f"(CPU ARI metric excluded), device RNG; random-N5C4S4S2-style scenes",
                       "global_batch": B * world, "per_gpu_batch": B, "parallelism": f"dp{world}"},
            "decoder_mfma_frac": round(ips / world * flop_img / (PEAK_MFMA_F32_TFLOPS * 1e12), 4), "final_loss": round(float(loss.item()), 4)}))
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=128, help="images per GPU")
    ap.add_argument("--obs-size", type=int, default=128)
    ap.add_argument("--num-slots", type=int, default=6, help="6 = headline; 16 with --obs-size 256 = BASELINE config 5")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["slate", "slotattn", "iodine"], default="slate",
                    help="slotattn = BASELINE config 2 (use_bcdec); iodine = config 4 (use --obs-size 64 --num-slots 7); headline = slate")
    ap.add_argument("--dropout", type=float, default=0.1, help="diagnostic only: the headline number uses the reference default 0.1")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from ocrl_amd import _lib, ocrs
    from ocrl_amd.utils.data import random_sprite_scenes, scenes_to_obs
    S, B = args.obs_size, args.batch
    if args.workload == "iodine":
        return bench_iodine(args, dev, dist, rank, world)
    ocr, env = slate_config(S, num_slots=args.num_slots)
    ocr.learning.dropout = args.dropout
    ocr.use_bcdec = args.workload == "slotattn"
    torch.manual_seed(0)                       # identical initial weights on every rank
    model = ocrs.SLATE(ocr, env)
    model._module._max_batch = B
    model.to(dev)
    model.train()
    model._module.set_seed(1 + rank)           # per-rank noise / dropout streams
    pool = [scenes_to_obs(random_sprite_scenes(B, S, seed=1000 * rank + i)).to(dev) for i in range(4)]

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()

    step = 0
    for _ in range(args.warmup):
        model.update(pool[step % len(pool)], None, step)
        step += 1
    L = _lib.lib()
    sync()
    L.ocrl_prof_enable(1 << 0)                 # time the conv5x5/64ch family on the launch stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        metrics = model.update(pool[step % len(pool)], None, step)
        step += 1
    sync()
    dt = time.perf_counter() - t0
    ms = (ctypes.c_double * 8)()
    cnt = (ctypes.c_longlong * 8)()
    _lib.check(L.ocrl_prof_collect(ctypes.byref(ms), ctypes.byref(cnt), 8))
    L.ocrl_prof_enable(0)
    loss = float(metrics["loss"].item())
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    ips = B * world * args.steps / dt
    # dominant kernel: conv_fwd_kernel<5,64,64>; algorithmic FLOPs per launch = 2 * (25*64) * 64 * B*S*S
    conv_flops = 2.0 * 25 * 64 * 64 * B * S * S
    conv_ms = ms[0] / max(cnt[0], 1)
    conv_tf = conv_flops / (conv_ms * 1e-3) / 1e12 if cnt[0] else 0.0
    traffic = None       # HBM bytes per launch from committed rocprofv3 PMC passes (same kernel, same shape), if present
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_step_traffic.json")))
        if B == 128 and S == 128:
            traffic = tj["kernels"]["void conv_fwd_kernel<5, 64, 64>(ConvArgs)"]["hbm_bytes_per_launch"]
    except Exception:
        pass
    out = {
        "metric": f"images/sec (node) SLATE pretrain {S}x{S}, {args.num_slots} slots, 3 iters" if args.workload == "slate" else
                  f"images/sec (node) Slot-Attention (use_bcdec) pretrain {S}x{S}, {args.num_slots} slots, 3 iters",
        "value": round(ips, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"SLATE {S}x{S}, {args.num_slots} slots, 3 iters, vocab 4096, d_model 192, 4 decoder blocks; full update() step "
                               f"(fwd+bwd+all-reduce+inf-norm clip+Adam), train mode dropout 0.1, device RNG; random-N5C4S4S2-style scenes",
                   "global_batch": B * world, "per_gpu_batch": B, "parallelism": f"dp{world}"},
        "roofline": {"bound": "mfma", "kernel": "conv_fwd_kernel<5,64,64> (CNN encoder 5x5 conv, fwd + bwd-data launches)",
                     "achieved": round(conv_tf, 2), "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(conv_tf / PEAK_MFMA_F32_TFLOPS, 4), "traffic": traffic,
                     "launches": int(cnt[0]), "avg_ms": round(conv_ms, 4)},
        "step_mfma_frac": round(ips / world * 3 * FWD_FLOP_PER_IMAGE.get((S, args.num_slots), 0) / (PEAK_MFMA_F32_TFLOPS * 1e12), 4),
        "final_loss": round(loss, 4),
    }
    if args.workload != "slate":
        out["config"]["workload"] = out["config"]["workload"].replace("SLATE", "Slot-Attention (use_bcdec) SLATE-encoder")
        out.pop("step_mfma_frac", None)
    if world == 1 and not args.no_cpu_baseline and args.workload == "slate":
        out["cpu_baseline"] = cpu_baseline(S, 4, 2)
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
