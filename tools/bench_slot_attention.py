"""Slot-attention loop micro-benchmark through the C ABI (ocrl_slot_attention_fwd/bwd): time and algorithmic HBM GB/s.
Algorithmic bytes (folded projections): forward I*N*64*4 per image; backward I*(x read + dx write) + (I-1)*dx re-read."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrl_amd import _lib
L = _lib.lib()
p = _lib.ptr
B, N, K, D, H, I = int(os.environ.get("B", 128)), int(os.environ.get("N", 16384)), int(os.environ.get("K", 6)), 192, 192, 3
C = 64
shp = [(C,), (C,), (D,), (D,), (D,), (D,), (D, D), (D, C), (D, C), (3 * D, D), (3 * D, D), (3 * D,), (3 * D,), (H, D), (H,), (D, H), (D,)]
w = [(torch.randn(s, device="cuda") / (s[-1] ** 0.5 if len(s) > 1 else 10.0) + (1.0 if len(s) == 1 and i in (0, 2, 4) else 0.0)) for i, s in enumerate(shp)]
g = [torch.zeros_like(t) for t in w]
x = torch.randn(B, N, C, device="cuda"); s0 = torch.randn(B, K, D, device="cuda"); ds = torch.randn(B, K, D, device="cuda")
slots = torch.empty(B, K, D, device="cuda"); attn = torch.empty(B, N, K, device="cuda"); dx = torch.empty_like(x); ds0 = torch.empty_like(s0)
nws = L.ocrl_slot_attention_ws_floats(B, K, D, H, I)
ws = torch.empty(nws, device="cuda")
arr = (ctypes.c_void_p * 17)(*[t.data_ptr() for t in w]); garr = (ctypes.c_void_p * 17)(*[t.data_ptr() for t in g])
def fwd(): _lib.check(L.ocrl_slot_attention_fwd(p(x), p(s0), arr, p(slots), p(attn), B, N, K, D, H, I, p(ws), nws, None))
def bwd(): _lib.check(L.ocrl_slot_attention_bwd(p(x), p(ds), p(dx), p(ds0), garr, B, N, K, D, H, I, p(ws), nws, None))
def t(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
bf = I * B * N * C * 4.0
bb = (I * 2 + (I - 1)) * B * N * C * 4.0
ms = t(fwd); print(f"slot_attn fwd B{B} N{N} K{K}: {ms:.3f} ms (incl. weight pack), {bf/ms/1e9:.2f} TB/s algorithmic")
ms = t(bwd); print(f"slot_attn bwd B{B} N{N} K{K}: {ms:.3f} ms (incl. 7 weight-gradient GEMMs), {bb/ms/1e9:.2f} TB/s algorithmic")
