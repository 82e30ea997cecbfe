"""Where does the encoder weight-gradient error of the 64x64 'long' parity case come from?
Compares, for the CNN encoder convolutions: the HIP path, the fp32 oracle and an fp64 run of the oracle; counts ReLU-mask
flips between the HIP activations and the fp64 pre-activations; and runs the conv weight-gradient / backward-data kernels on
the oracle's own operands (rounded to fp32) against fp64, which isolates their summation error from everything upstream.

    python tests/diag_encoder_grad.py            (GPU box)
"""
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import slate_oracle as O          # noqa: E402  (test-side diagnostic: the oracle is the checker)
from tests.gpu_util import dims_from_cfg, load_params, relerr   # noqa: E402
from tests.test_gpu_slate import LONG, dev_noise               # noqa: E402
from ocrl_amd import _lib                                       # noqa: E402
from ocrl_amd.engine import SlateEngine                         # noqa: E402

PRE = {}
FORCED_MASKS = None       # per layer [B,64,S,S] bool: use these ReLU masks instead of (pre > 0)


def cnn_encode_hooked(P, obs, grid=None):
    x = obs
    for i in range(3):
        pre = F.conv2d(x, P[f"_enc._encoder.{i}.m.weight"], P[f"_enc._encoder.{i}.m.bias"], padding=2)
        pre.retain_grad()
        PRE[i] = (x, pre)
        x = F.relu(pre) if FORCED_MASKS is None else pre * FORCED_MASKS[i].to(pre.dtype)
    pre = F.conv2d(x, P["_enc._encoder.3.weight"], P["_enc._encoder.3.bias"], padding=2)
    pre.retain_grad()
    PRE[3] = (x, pre)
    grid = O.position_grid(obs.shape[-1]).to(pre.dtype)
    x = pre + F.conv2d(grid, P["_enc_pos.channels_map.weight"], P["_enc_pos.channels_map.bias"])
    return x.permute(0, 2, 3, 1).flatten(1, 2)


def run_oracle(cfg, P, obs, noise, step, dtype):
    if dtype == torch.float64:
        P = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
        obs, noise = obs.double(), {k: v.double() for k, v in noise.items()}
    tr = O.OracleTrainer(cfg, P)
    res = tr.loss_and_grads(obs, noise, step, None)
    return tr, res


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def main():
    torch.set_num_threads(16)
    tag = sys.argv[1] if len(sys.argv) > 1 else "long"
    cfg = O.default_cfg(**LONG)
    B, S = 2, cfg.obs_size
    P = O.formula_params(cfg)
    obs = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(100))
    noise = O.make_noise(cfg, B, 7)
    step = 10
    tau, _ = O.schedules(cfg, step)
    t32, _ = run_oracle(cfg, P, obs, noise, step, torch.float32)
    O.cnn_encode = cnn_encode_hooked
    t64, _ = run_oracle(cfg, P, obs, noise, step, torch.float64)
    eng = SlateEngine(dims_from_cfg(cfg), max_batch=B)
    load_params(eng, P)
    eng.forward(obs.cuda(), tau, train=False, seed=1, noise=dev_noise(cfg, noise))
    eng.backward()
    torch.cuda.synchronize()
    gmax = max(t64.P[p.name].grad.abs().max().item() for p in eng.params)
    print(f"[{tag}] per-tensor max-norm error vs the fp64 oracle (floor 1e-5 x largest gradient)")
    for p in eng.params:
        if not p.name.startswith("_enc.") and "slotattn.mlp" not in p.name:
            continue
        ref = t64.P[p.name].grad.reshape(p.shape)
        e_hip = relerr(eng.view(eng.flat_g, p), ref, floor=1e-5 * gmax)
        e_o32 = relerr(t32.P[p.name].grad.reshape(p.shape), ref, floor=1e-5 * gmax)
        print(f"  {p.name:34s} HIP {e_hip:.2e}   fp32 oracle {e_o32:.2e}")
    # ReLU mask flips: HIP activation > 0 vs fp64 pre-activation > 0
    for i, name in enumerate(("enc1", "enc2", "enc3")):
        act = eng.tensor(name, (B, S, S, 64)).cpu()
        pre = nhwc(PRE[i][1].detach())
        flips = ((act > 0) != (pre > 0))
        print(f"  layer {i}: ReLU mask flips HIP vs fp64: {int(flips.sum())} of {flips.numel()}; smallest |pre| {pre.abs().min().item():.2e}; "
              f"|pre| at flips {pre[flips].abs().tolist()[:8]}; activation error {relerr(act, F.relu(pre)):.2e}")
    # the same fp64 run with the ReLU masks of the HIP forward: if mask flips are the whole story, the error collapses
    global FORCED_MASKS
    FORCED_MASKS = [(eng.tensor(name, (B, S, S, 64)).cpu() > 0).permute(0, 3, 1, 2) for name in ("enc1", "enc2", "enc3")]
    t64m, _ = run_oracle(cfg, P, obs, noise, step, torch.float64)
    FORCED_MASKS = None
    print(f"[{tag}] against an fp64 run that uses the HIP forward's ReLU masks in the CNN encoder:")
    for p in eng.params:
        if p.name.startswith("_enc."):
            ref = t64m.P[p.name].grad.reshape(p.shape)
            print(f"  {p.name:34s} HIP {relerr(eng.view(eng.flat_g, p), ref, floor=1e-5 * gmax):.2e}")
    # unit kernels on the oracle's operands
    L = _lib.lib()
    for i in (1, 2, 3):
        x = PRE[i][0].detach().float()
        dy = PRE[i][1].grad.float()
        w64 = torch.zeros(64, 64, 5, 5, dtype=torch.double, requires_grad=True)
        F.conv2d(x.double(), w64, None, padding=2).backward(dy.double())
        n = L.ocrl_conv2d_wgrad_ws_floats(B, S, S, 5, 64)
        ws = torch.empty(n, device="cuda")
        dw = torch.zeros(64, 64, 5, 5, device="cuda")
        db = torch.zeros(64, device="cuda")
        xd, dyd = nhwc(x).cuda(), nhwc(dy).cuda()
        _lib.check(L.ocrl_conv2d_bwd_weight(_lib.ptr(xd), _lib.ptr(dyd), _lib.ptr(dw), _lib.ptr(db), B, S, S, 64, 64, 5, _lib.ptr(ws), n, None))
        torch.cuda.synchronize()
        w32 = torch.zeros(64, 64, 5, 5, requires_grad=True)
        F.conv2d(x, w32, None, padding=2).backward(dy)
        absum = F.conv2d(x.double().abs().transpose(0, 1), dy.double().abs().transpose(0, 1), padding=2).transpose(0, 1)   # sum |x||dy| per weight
        print(f"  layer {i} wgrad on the oracle's operands: HIP {relerr(dw.cpu(), w64.grad):.2e}, torch fp32 {relerr(w32.grad, w64.grad):.2e} of max |dW| "
              f"{w64.grad.abs().max().item():.3e};  max |dW| / max sum|x||dy| = {w64.grad.abs().max().item() / absum.max().item():.2e} (cancellation)")
        full = relerr(eng.grad(f"_enc._encoder.{i}.m.weight" if i < 3 else "_enc._encoder.3.weight"), t64.P[f"_enc._encoder.{i}.m.weight" if i < 3 else "_enc._encoder.3.weight"].grad)
        print(f"           full path, same tensor: {full:.2e}")


if __name__ == "__main__":
    main()
