"""attention micro-benchmark through the C ABI (development aid)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrl_amd import _lib
L = _lib.lib()
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
B, T, d, h = 128, 1024, 192, 4
p = float(os.environ.get("P", "0.1"))
qkv = torch.randn(B, T, 3 * d, device="cuda")
q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
o = torch.empty(B, T, d, device="cuda"); lse = torch.empty(B, h, T, device="cuda")
dO = torch.randn(B, T, d, device="cuda"); dqkv = torch.empty_like(qkv); delta = torch.empty(B, h, T, device="cuda")
def fwd(): _lib.check(L.ocrl_attention_fwd(P(q), P(k), P(v), P(o), P(lse), B, T, d, h, 3 * d, p, 1, 16, None))
def bwd(): _lib.check(L.ocrl_attention_bwd(P(q), P(k), P(v), P(o), P(lse), P(dO), P(dqkv[..., :d]), P(dqkv[..., d:2*d]), P(dqkv[..., 2*d:]), P(delta), B, T, d, h, 3 * d, p, 1, 16, None))
def t(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
fl = 4.0 * B * h * T * T * (d // h) / 2       # causal: half of 2 matmuls x 2 flop
ms = t(fwd); print(f"attn fwd B{B} T{T} p={p}: {ms:.3f} ms {fl/ms/1e9:.1f} TFLOP/s (causal flops)")
ms = t(bwd); print(f"attn bwd B{B} T{T} p={p}: {ms:.3f} ms {3.5*fl/ms/1e9:.1f} TFLOP/s (7 products)")
