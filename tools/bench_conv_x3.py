"""Exploratory: the 5x5 / 64-channel convolution at the benched shape (B=128, 128x128) on the fp32 MFMA kernel and on the split-precision
bf16 kernel (csrc/conv_x3.hip), forward and backward-data: ms per launch and fp32-equivalent TFLOP/s (2*25*64*64 FLOP per output pixel)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrl_amd import _lib
L = _lib.lib(); P = _lib.ptr
B, S, KS = int(os.environ.get("B", 128)), int(os.environ.get("S", 128)), int(os.environ.get("KS", 5))
x = torch.randn(B, S, S, 64, device="cuda"); w = torch.randn(64, 64, KS, KS, device="cuda") / (8 * KS); b = torch.randn(64, device="cuda")
if os.environ.get("ZERO"):          # DVFS check: the same instruction stream on zeros draws less power and holds a higher clock
    x.zero_(); w.zero_()
y = torch.empty_like(x); act = torch.randn_like(x)
ws3 = torch.empty(L.ocrl_conv2d_x3_ws_floats(), device="cuda"); ws = torch.empty(2 * 25 * 64 * 64, device="cuda")
nw = L.ocrl_conv2d_wgrad_ws_floats(B, S, S, KS, 64); wsw = torch.empty(nw, device="cuda"); dw = torch.empty(64, 64, KS, KS, device="cuda")
flop = 2.0 * KS * KS * 64 * 64 * B * S * S
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, f in (("fp32 MFMA forward", lambda: _lib.check(L.ocrl_conv2d_fwd(P(x), P(w), P(b), P(y), B, S, S, 64, 64, KS, 1, P(ws), None))),
                ("3xbf16 split forward", lambda: _lib.check(L.ocrl_conv2d_fwd_x3(P(x), P(w), P(b), P(y), B, S, S, KS, 1, P(ws3), None))),
                ("fp32 MFMA backward-data", lambda: _lib.check(L.ocrl_conv2d_bwd_data(P(x), P(w), P(act), P(y), B, S, S, KS, P(ws), None))),
                ("3xbf16 split backward-data", lambda: _lib.check(L.ocrl_conv2d_bwd_data_x3(P(x), P(w), P(act), P(y), B, S, S, KS, P(ws3), None))),
                ("fp32 MFMA weight gradient", lambda: _lib.check(L.ocrl_conv2d_bwd_weight(P(x), P(act), P(dw), None, B, S, S, 64, 64, KS, P(wsw), nw, None))),
                ("3xbf16 split weight gradient", lambda: _lib.check(L.ocrl_conv2d_bwd_weight_x3(P(x), P(act), P(dw), B, S, S, KS, P(wsw), nw, None)))):
    ms = t(f)
    print(f"{name:28s} {KS}x{KS} B{B} {S}x{S}: {ms:.3f} ms (incl. the weight pack launch), {flop / ms / 1e9:.1f} TFLOP/s fp32-equivalent", flush=True)
