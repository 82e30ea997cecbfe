"""What the vendor library (rocBLAS / hipBLASLt through torch.matmul, fp32) reaches on the step's GEMM shapes — a yardstick for
gemm.hip, not part of the product path."""
import torch
torch.backends.cuda.matmul.allow_tf32 = False
SHAPES = [("proj NT", 131072, 192, 192, "nt"), ("qkv NT", 131072, 576, 192, "nt"), ("ffn1 NT", 131072, 768, 192, "nt"), ("ffn2dx NN", 131072, 768, 192, "nn"), ("projdx NN", 131072, 192, 192, "nn"), ("ffn2 NT", 131072, 192, 768, "nt"), ("head NT", 131072, 4096, 192, "nt"),
          ("headdx NN", 131072, 192, 4096, "nn"), ("samlp NT", 2097152, 64, 64, "nt"), ("dW TN 192", 192, 192, 131072, "tn"), ("dW TN 768x192", 768, 192, 131072, "tn"),
          ("dW TN 4096x192", 4096, 192, 131072, "tn")]
for name, M, N, K, form in SHAPES:
    if form == "nt":
        A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); f = lambda: A @ B.t()
    elif form == "nn":
        A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda"); f = lambda: A @ B
    else:
        A = torch.randn(K, M, device="cuda"); B = torch.randn(K, N, device="cuda"); f = lambda: A.t() @ B
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name:18s} {M}x{N}x{K}: {ms*1e3:9.1f} us {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)
