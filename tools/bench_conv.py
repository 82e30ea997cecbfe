"""conv micro-benchmark through the C ABI (development aid)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrl_amd import _lib
L = _lib.lib()
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
B, S = 128, 128
x = torch.randn(B, S, S, 64, device="cuda")
w = torch.randn(64, 64, 5, 5, device="cuda") * 0.02
b = torch.zeros(64, device="cuda")
y = torch.empty(B, S, S, 64, device="cuda")
ws = torch.empty(2 * 25 * 64 * 64, device="cuda")
fl = 2.0 * 25 * 64 * 64 * B * S * S
def t(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ms = t(lambda: _lib.check(L.ocrl_conv2d_fwd(P(x), P(w), P(b), P(y), B, S, S, 64, 64, 5, 1, P(ws), None)))
print(f"conv fwd 5x5 64->64 B{B} S{S}: {ms:.3f} ms {fl/ms/1e9:.1f} TFLOP/s")
n = L.ocrl_conv2d_wgrad_ws_floats(B, S, S, 5, 64)
ws2 = torch.empty(n, device="cuda"); dw = torch.empty(64, 64, 5, 5, device="cuda")
ms = t(lambda: _lib.check(L.ocrl_conv2d_bwd_weight(P(x), P(y), P(dw), None, B, S, S, 64, 64, 5, P(ws2), n, None)))
print(f"conv wgrad 5x5 64->64 B{B} S{S}: {ms:.3f} ms {fl/ms/1e9:.1f} TFLOP/s")
# 3x3 (IODINE decoder shape: B*K = 896 maps of 64x64)
B3, S3 = 896, 64
x3 = torch.randn(B3, S3, S3, 64, device="cuda"); w3 = torch.randn(64, 64, 3, 3, device="cuda") * 0.03; y3 = torch.empty_like(x3)
fl3 = 2.0 * 9 * 64 * 64 * B3 * S3 * S3
ms = t(lambda: _lib.check(L.ocrl_conv2d_fwd(P(x3), P(w3), P(b), P(y3), B3, S3, S3, 64, 64, 3, 1, P(ws), None)))
print(f"conv fwd 3x3 64->64 B{B3} S{S3}: {ms:.3f} ms {fl3/ms/1e9:.1f} TFLOP/s")
