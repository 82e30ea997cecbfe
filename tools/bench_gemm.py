"""GEMM micro-benchmark through the C ABI (development aid): prints TFLOP/s per shape.
   OCRL_GEMM_TILE=128x64 python tools/bench_gemm.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrl_amd import _lib

L = _lib.lib()
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
SHAPES = [  # (name, M, N, K, akc, bkc, splitk)
    ("proj NT 131072x192x192", 131072, 192, 192, 1, 1, 1),
    ("proj NN 131072x192x192", 131072, 192, 192, 1, 0, 1),
    ("qkv NT 131072x576x192", 131072, 576, 192, 1, 1, 1),
    ("ffn1 NT 131072x768x192", 131072, 768, 192, 1, 1, 1),
    ("ffn2 NT 131072x192x768", 131072, 192, 768, 1, 1, 1),
    ("ffn2dx NN 131072x768x192", 131072, 768, 192, 1, 0, 1),
    ("head NT 131072x4096x192", 131072, 4096, 192, 1, 1, 1),
    ("headdx NN 131072x192x4096", 131072, 192, 4096, 1, 0, 1),
    ("samlp NT 2097152x64x64", 2097152, 64, 64, 1, 1, 1),
    ("samlpdx NN 2097152x64x64", 2097152, 64, 64, 1, 0, 1),
    ("dW TN 192x192x131072 s256", 192, 192, 131072, 0, 0, 256),
    ("dW TN 768x192x131072 s85", 768, 192, 131072, 0, 0, 85),
    ("dW TN 64x64x2097152 s1024", 64, 64, 2097152, 0, 0, 1024),
    ("dW TN 4096x192x131072 s16", 4096, 192, 131072, 0, 0, 16),
    # the skinny weight gradients of the dVAE backward (side stream): decoder.9 [256,64] over 4BT rows, decoder.11 [4,64] over BN rows,
    # encoder.7 [4096,64] over BT rows, the 64x64 layers over BT / 4BT rows
    ("dvae dW TN 256x64x524288 s512", 256, 64, 524288, 0, 0, 512),
    ("dvae dW TN 4x64x2097152 s512", 4, 64, 2097152, 0, 0, 512),
    ("dvae dW TN 4096x64x131072 s32", 4096, 64, 131072, 0, 0, 32),
    ("dvae dW TN 64x64x524288 s1024", 64, 64, 524288, 0, 0, 1024),
    ("dvae dX NN 524288x64x256", 524288, 64, 256, 1, 0, 1),
    ("dvae dX NN 131072x64x4096", 131072, 64, 4096, 1, 0, 1),
]
only = sys.argv[1:] 
for name, M, N, K, akc, bkc, sk in SHAPES:
    if only and not any(o in name for o in only):
        continue
    A = torch.randn((M, K) if akc else (K, M), device="cuda")
    B = torch.randn((N, K) if bkc else (K, N), device="cuda")
    C = torch.empty(M, N, device="cuda")
    ws = torch.empty(sk * M * N, device="cuda") if sk > 1 else None
    lda, ldb = A.shape[1], B.shape[1]
    def run():
        _lib.check(L.ocrl_gemm(P(A), P(B), P(C), M, N, K, lda, ldb, N, akc, bkc, 1.0, None, 0, None, 0, None, 0, sk, P(ws), None))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name:32s} {ms*1e3:9.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s  tile={os.environ.get('OCRL_GEMM_TILE','auto')}", flush=True)
