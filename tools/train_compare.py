"""Training-dynamics check of the exploratory split-precision convolutions: the same `train_ocr.py ocr=slate` run (64x64 scenes, batch 64,
device RNG, same seeds) once on the fp32-MFMA convolutions, once with OCRL_CONV_X3=1 and, as the control for how far two fp32 runs
drift apart by summation order alone, once with the unfused fp32 cross-attention kernels (OCRL_XATTN=0), each in its own process; prints
the loss every 50 steps side by side and writes gpurun_out/train_compare.json."""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = int(os.environ.get("STEPS", 300))
out = {}
for tag, env in (("fp32_mfma", {}), ("split_precision_x3", {"OCRL_CONV_X3": "1"}), ("fp32_mfma_other_summation_order", {"OCRL_XATTN": "0"})):
    run = tempfile.mkdtemp(prefix="tc_" + tag)
    args = ["ocr=slate", "ocr.slotattr.num_slots=6", "ocr.slotattr.num_iterations=3", "dataset=random-N5C4S4S2", "device=cuda:0", "batch_size=64",
            "num_workers=0", "dataset.synthetic_train=1024", "dataset.synthetic_val=64", "eval_interval=100000", f"max_steps={STEPS}", "log_interval=1", f"run_dir={run}"]
    code = f"import sys; sys.path.insert(0, {ROOT!r}); import train_ocr; train_ocr.main({args!r})"
    subprocess.run([sys.executable, "-c", code], check=True, env={**os.environ, **env}, stdout=subprocess.DEVNULL)
    lines = [json.loads(l) for l in open(os.path.join(run, "metrics.jsonl"))]
    out[tag] = [(l["step"] if "step" in l else i, l["train/loss"], l["train/dvae_mse"], l["train/cross_entropy"]) for i, l in enumerate(lines) if "train/loss" in l]
a, b, c = out["fp32_mfma"], out["split_precision_x3"], out["fp32_mfma_other_summation_order"]
print(f"{'step':>5} {'loss fp32-MFMA':>16} {'loss split x3':>16} {'rel diff':>10} {'fp32, other order':>18} {'rel diff':>10}")
for i in list(range(0, len(a), 50)) + [len(a) - 1]:
    print(f"{i:5d} {a[i][1]:16.4f} {b[i][1]:16.4f} {abs(a[i][1] - b[i][1]) / abs(a[i][1]):10.2e} {c[i][1]:18.4f} {abs(a[i][1] - c[i][1]) / abs(a[i][1]):10.2e}")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"steps": STEPS, "config": "train_ocr.py ocr=slate 64x64, 6 slots, batch 64, synthetic scenes, device RNG, identical seeds",
           "loss_every_10_steps": {k: [v[i][1] for i in range(0, len(v), 10)] for k, v in out.items()}}, open(os.path.join(ROOT, "gpurun_out", "train_compare.json"), "w"))
