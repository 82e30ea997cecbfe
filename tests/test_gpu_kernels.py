"""Unit parity of the MFMA kernels through the C ABI against plain PyTorch fp32 (CPU, float64-accumulated
where noted).  Tolerance: 2e-5 relative to the output's max magnitude (fp32, different summation order)."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from tests.gpu_util import log, relerr

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module")
def L():
    from ocrl_amd import _lib
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return _lib


def dev(t):
    return t.to("cuda").contiguous()


def P(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def run_gemm(L, A, B, M, N, K, akc, bkc, alpha=1.0, bias=None, relu=0, mask=None, resid=None, splitk=1, ldc=None):
    ldc = ldc or N
    C = torch.zeros(M, ldc, device="cuda")
    ws = torch.empty(max(1, splitk * M * N), device="cuda") if splitk > 1 else None
    lda = A.shape[1]
    ldb = B.shape[1]
    L.check(L.lib().ocrl_gemm(P(A), P(B), P(C), M, N, K, lda, ldb, ldc, akc, bkc, alpha, P(bias), relu, P(mask),
                               0 if mask is None else mask.shape[1], P(resid), 0 if resid is None else resid.shape[1], splitk, P(ws), None))
    torch.cuda.synchronize()
    return C[:, :N].cpu()


@pytest.mark.parametrize("M,N,K", [(300, 64, 48), (1000, 192, 192), (257, 3, 64), (512, 4096, 64), (130, 768, 192), (64, 64, 4096), (12, 192, 192)])
def test_gemm_nt(L, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    R = torch.randn(M, N if N % 4 == 0 else 4, generator=g)
    ref = torch.relu(A.double() @ W.double().T + b.double())
    out = run_gemm(L, dev(A), dev(W), M, N, K, 1, 1, bias=dev(b), relu=1, ldc=N if N % 4 == 0 else 4)
    e = relerr(out, ref)
    log(f"gemm NT bias+relu {M}x{N}x{K}: {e:.2e}")
    assert e < TOL
    if N % 4 == 0:
        ref2 = 0.5 * (A.double() @ W.double().T) + R.double()
        out2 = run_gemm(L, dev(A), dev(W), M, N, K, 1, 1, alpha=0.5, resid=dev(R))
        e2 = relerr(out2, ref2)
        log(f"gemm NT alpha+resid {M}x{N}x{K}: {e2:.2e}")
        assert e2 < TOL


@pytest.mark.parametrize("M,Nout,Kin", [(500, 192, 768), (333, 4096, 192), (1000, 64, 64), (200, 4, 64)])
def test_gemm_nn_dx(L, M, Nout, Kin):
    g = torch.Generator().manual_seed(M + Nout)
    dY = torch.randn(M, Nout, generator=g)
    W = torch.randn(Nout, Kin, generator=g) / Nout ** 0.5
    act = torch.randn(M, Kin, generator=g)
    ref = (dY.double() @ W.double()) * (act > 0)
    out = run_gemm(L, dev(dY), dev(W), M, Kin, Nout, 1, 0, mask=dev(act))
    e = relerr(out, ref)
    log(f"gemm NN dx+mask {M}x{Nout}x{Kin}: {e:.2e}")
    assert e < TOL


@pytest.mark.parametrize("M,Nout,Kin,splitk", [(5000, 64, 64, 8), (3000, 192, 768, 1), (2048, 4096, 192, 4), (4100, 4, 64, 4), (12, 192, 192, 1), (777, 64, 48, 2)])
def test_gemm_tn_dw(L, M, Nout, Kin, splitk):
    g = torch.Generator().manual_seed(M + Kin)
    dY = torch.randn(M, Nout, generator=g)
    X = torch.randn(M, Kin, generator=g)
    ref = dY.double().T @ X.double()
    out = run_gemm(L, dev(dY), dev(X), Nout, Kin, M, 0, 0, splitk=splitk)
    e = relerr(out, ref)
    log(f"gemm TN dW {M}x{Nout}x{Kin} splitk={splitk}: {e:.2e}")
    assert e < TOL


def nhwc(x, cpad=None):
    x = x.permute(0, 2, 3, 1).contiguous()
    if cpad and cpad > x.shape[-1]:
        x = F.pad(x, (0, cpad - x.shape[-1]))
    return x.contiguous()


@pytest.mark.parametrize("B,S,cin,ks", [(2, 16, 64, 5), (2, 32, 64, 5), (3, 20, 64, 3), (2, 16, 3, 5), (1, 64, 3, 5), (2, 8, 64, 3)])
def test_conv_fwd(L, B, S, cin, ks):
    g = torch.Generator().manual_seed(B * S + cin)
    x = torch.randn(B, cin, S, S, generator=g)
    w = torch.randn(64, cin, ks, ks, generator=g) / (cin * ks * ks) ** 0.5
    b = torch.randn(64, generator=g)
    ref = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=ks // 2))
    cpad = 8 if cin < 8 else 64
    xd, wd, bd = dev(nhwc(x, cpad)), dev(w), dev(b)        # keep references: the launch is asynchronous
    y = torch.empty(B, S, S, 64, device="cuda")
    ws = torch.empty(ks * ks * cpad * 64, device="cuda")
    L.check(L.lib().ocrl_conv2d_fwd(P(xd), P(wd), P(bd), P(y), B, S, S, cin, cpad, ks, 1, P(ws), None))
    torch.cuda.synchronize()
    e = relerr(y.cpu().permute(0, 3, 1, 2), ref)
    log(f"conv fwd B{B} S{S} cin{cin} ks{ks}: {e:.2e}")
    assert e < TOL


@pytest.mark.parametrize("B,H,W,relu", [(1, 64, 64, 1), (2, 16, 16, 1), (3, 20, 44, 0), (1, 128, 128, 1), (8, 64, 64, 1), (1, 7, 33, 1)])
def test_conv_fwd_low_latency_variant(L, B, H, W, relu):
    """ocrl_conv2d_fwd_lowlat: the k-split kernel the inference path (encode at a few images) uses for the 5x5 / 64-channel layers --
    ragged widths (a partial 32-pixel segment), heights that are no multiple of the throughput kernel's 4-row tile, with and without
    ReLU -- against F.conv2d in fp64 and against the throughput kernel (same arithmetic, other summation order)"""
    g = torch.Generator().manual_seed(B * H + W)
    x = torch.randn(B, 64, H, W, generator=g)
    w = torch.randn(64, 64, 5, 5, generator=g) / 40.0
    b = torch.randn(64, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=2)
    if relu:
        ref = torch.relu(ref)
    xd, wd, bd = dev(nhwc(x)), dev(w), dev(b)
    y, y0 = torch.full((B, H, W, 64), float("nan"), device="cuda"), torch.empty(B, H, W, 64, device="cuda")
    ws = torch.empty(25 * 64 * 64, device="cuda")
    L.check(L.lib().ocrl_conv2d_fwd_lowlat(P(xd), P(wd), P(bd), P(y), B, H, W, 64, 64, 5, relu, P(ws), None))
    L.check(L.lib().ocrl_conv2d_fwd(P(xd), P(wd), P(bd), P(y0), B, H, W, 64, 64, 5, relu, P(ws), None))
    torch.cuda.synchronize()
    e, e0 = relerr(y.cpu().permute(0, 3, 1, 2), ref), relerr(y.cpu(), y0.cpu())
    log(f"conv fwd low-latency B{B} {H}x{W} relu{relu}: vs fp64 {e:.2e}, vs the throughput kernel {e0:.2e}")
    assert torch.isfinite(y).all() and e < TOL and e0 < TOL


@pytest.mark.parametrize("B,H,W,ks", [(2, 16, 16, 5), (1, 64, 64, 5), (3, 21, 44, 5), (2, 128, 128, 5), (2, 16, 16, 3), (3, 21, 44, 3), (2, 64, 64, 3)])
def test_conv_split_precision_exploratory(L, B, H, W, ks):
    """csrc/conv_x3.hip (exploratory, not on a default path): the 5x5 (3x3) / 64-channel layer on the bf16 matrix pipe with every fp32 operand
    split exactly into three bf16 numbers, six products accumulated in fp32 -- forward (bias + ReLU) and backward-data (ReLU mask) must
    hold the fp32 kernels' tolerance against fp64, i.e. it is fp32-equivalent arithmetic, not bf16 arithmetic"""
    g = torch.Generator().manual_seed(B * H + W)
    x = torch.randn(B, 64, H, W, generator=g) * torch.exp(2.0 * torch.randn(B, 64, H, W, generator=g))     # magnitudes over several binades
    w = torch.randn(64, 64, ks, ks, generator=g) / (8.0 * ks)
    b = torch.randn(64, generator=g)
    ref = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=ks // 2))
    xd, wd, bd = dev(nhwc(x)), dev(w), dev(b)
    y, y0 = torch.full((B, H, W, 64), float("nan"), device="cuda"), torch.empty(B, H, W, 64, device="cuda")
    ws = torch.empty(L.lib().ocrl_conv2d_x3_ws_floats(), device="cuda")
    ws0 = torch.empty(2 * 25 * 64 * 64, device="cuda")
    L.check(L.lib().ocrl_conv2d_fwd_x3(P(xd), P(wd), P(bd), P(y), B, H, W, ks, 1, P(ws), None))
    L.check(L.lib().ocrl_conv2d_fwd(P(xd), P(wd), P(bd), P(y0), B, H, W, 64, 64, ks, 1, P(ws0), None))
    torch.cuda.synchronize()
    e, e0 = relerr(y.cpu().permute(0, 3, 1, 2), ref), relerr(y0.cpu().permute(0, 3, 1, 2), ref)
    # backward data with a ReLU mask
    dy = torch.randn(B, 64, H, W, generator=g)
    act = torch.randn(B, 64, H, W, generator=g)
    xg = torch.zeros(B, 64, H, W, dtype=torch.double, requires_grad=True)
    F.conv2d(xg, w.double(), None, padding=ks // 2).backward(dy.double())
    refb = xg.grad * (act > 0)
    dyd, actd = dev(nhwc(dy)), dev(nhwc(act))
    dx, dx0 = torch.full((B, H, W, 64), float("nan"), device="cuda"), torch.empty(B, H, W, 64, device="cuda")
    L.check(L.lib().ocrl_conv2d_bwd_data_x3(P(dyd), P(wd), P(actd), P(dx), B, H, W, ks, P(ws), None))
    L.check(L.lib().ocrl_conv2d_bwd_data(P(dyd), P(wd), P(actd), P(dx0), B, H, W, ks, P(ws0), None))
    torch.cuda.synchronize()
    eb, eb0 = relerr(dx.cpu().permute(0, 3, 1, 2), refb), relerr(dx0.cpu().permute(0, 3, 1, 2), refb)
    # weight gradient
    wg = torch.zeros(64, 64, ks, ks, dtype=torch.double, requires_grad=True)
    F.conv2d(x.double(), wg, None, padding=ks // 2).backward(dy.double())
    n = L.lib().ocrl_conv2d_wgrad_ws_floats(B, H, W, ks, 64)
    wsw = torch.empty(n, device="cuda")
    dw, dw0 = torch.full((64, 64, ks, ks), float("nan"), device="cuda"), torch.empty(64, 64, ks, ks, device="cuda")
    L.check(L.lib().ocrl_conv2d_bwd_weight_x3(P(xd), P(dyd), P(dw), B, H, W, ks, P(wsw), n, None))
    L.check(L.lib().ocrl_conv2d_bwd_weight(P(xd), P(dyd), P(dw0), None, B, H, W, 64, 64, ks, P(wsw), n, None))
    torch.cuda.synchronize()
    ew, ew0 = relerr(dw.cpu(), wg.grad), relerr(dw0.cpu(), wg.grad)
    log(f"conv {ks}x{ks} split-precision (3 x bf16, 6 products) B{B} {H}x{W}: forward {e:.2e} (fp32 MFMA kernel {e0:.2e}), backward-data {eb:.2e} ({eb0:.2e}), "
        f"weight gradient {ew:.2e} ({ew0:.2e}) vs fp64")
    assert torch.isfinite(y).all() and torch.isfinite(dx).all() and torch.isfinite(dw).all()
    assert e < TOL and eb < TOL and ew < TOL


@pytest.mark.parametrize("B,S,ks", [(2, 16, 5), (2, 32, 5), (2, 12, 3)])
def test_conv_bwd_data(L, B, S, ks):
    g = torch.Generator().manual_seed(B * S + ks)
    x = torch.randn(B, 64, S, S, generator=g).double().requires_grad_(True)
    w = (torch.randn(64, 64, ks, ks, generator=g) / (64 * ks * ks) ** 0.5)
    dy = torch.randn(B, 64, S, S, generator=g)
    act = torch.randn(B, 64, S, S, generator=g)
    F.conv2d(x, w.double(), None, padding=ks // 2).backward(dy.double())
    ref = x.grad * (act > 0)
    dx = torch.empty(B, S, S, 64, device="cuda")
    ws = torch.empty(2 * ks * ks * 64 * 64, device="cuda")
    dyd, wd, actd = dev(nhwc(dy)), dev(w), dev(nhwc(act))
    L.check(L.lib().ocrl_conv2d_bwd_data(P(dyd), P(wd), P(actd), P(dx), B, S, S, ks, P(ws), None))
    torch.cuda.synchronize()
    e = relerr(dx.cpu().permute(0, 3, 1, 2), ref)
    log(f"conv bwd-data B{B} S{S} ks{ks}: {e:.2e}")
    assert e < TOL


@pytest.mark.parametrize("B,S,cin,ks", [(2, 16, 64, 5), (3, 32, 64, 5), (2, 16, 3, 5), (2, 64, 3, 5), (2, 12, 64, 3), (5, 32, 64, 3)])
def test_conv_bwd_weight(L, B, S, cin, ks):
    g = torch.Generator().manual_seed(B * S + cin + ks)
    x = torch.randn(B, cin, S, S, generator=g)
    w = torch.zeros(64, cin, ks, ks, dtype=torch.double, requires_grad=True)
    bias = torch.zeros(64, dtype=torch.double, requires_grad=True)
    dy = torch.randn(B, 64, S, S, generator=g)
    F.conv2d(x.double(), w, bias, padding=ks // 2).backward(dy.double())
    cpad = 8 if cin < 8 else 64
    n = L.lib().ocrl_conv2d_wgrad_ws_floats(B, S, S, ks, cpad)
    ws = torch.empty(n, device="cuda")
    dw = torch.zeros(64, cin, ks, ks, device="cuda")
    db = torch.zeros(64, device="cuda")
    xd, dyd = dev(nhwc(x, cpad)), dev(nhwc(dy))
    L.check(L.lib().ocrl_conv2d_bwd_weight(P(xd), P(dyd), P(dw), P(db), B, S, S, cin, cpad, ks, P(ws), n, None))
    torch.cuda.synchronize()
    e, eb = relerr(dw.cpu(), w.grad), relerr(db.cpu(), bias.grad)
    log(f"conv wgrad B{B} S{S} cin{cin} ks{ks}: dW {e:.2e} db {eb:.2e}")
    assert e < TOL and eb < TOL


@pytest.mark.parametrize("R,Fd", [(1000, 64), (517, 192), (12, 192), (4096, 128)])
def test_layernorm(L, R, Fd):
    g = torch.Generator().manual_seed(R + Fd)
    x = (torch.randn(R, Fd, generator=g) * 2 + 0.5).double().requires_grad_(True)
    gam = (1 + 0.2 * torch.randn(Fd, generator=g)).double().requires_grad_(True)
    bet = (0.1 * torch.randn(Fd, generator=g)).double().requires_grad_(True)
    dy = torch.randn(R, Fd, generator=g)
    y = F.layer_norm(x, (Fd,), gam, bet)
    y.backward(dy.double())
    xd, gd, bd, dyd = dev(x.detach().float()), dev(gam.detach().float()), dev(bet.detach().float()), dev(dy)
    yo = torch.empty(R, Fd, device="cuda")
    mean = torch.empty(R, device="cuda")
    rstd = torch.empty(R, device="cuda")
    L.check(L.lib().ocrl_layernorm_fwd(P(xd), P(gd), P(bd), P(yo), P(mean), P(rstd), R, Fd, None))
    dx = torch.empty(R, Fd, device="cuda")
    dgb = torch.empty(2 * Fd, device="cuda")
    ws = torch.empty(1 << 20, device="cuda")
    L.check(L.lib().ocrl_layernorm_bwd(P(dyd), P(xd), P(mean), P(rstd), P(gd), P(dx), P(dgb), R, Fd, P(ws), ws.numel(), None))
    torch.cuda.synchronize()
    e = [relerr(yo.cpu(), y), relerr(dx.cpu(), x.grad), relerr(dgb[:Fd].cpu(), gam.grad), relerr(dgb[Fd:].cpu(), bet.grad)]
    log(f"layernorm R{R} F{Fd}: y {e[0]:.2e} dx {e[1]:.2e} dgamma {e[2]:.2e} dbeta {e[3]:.2e}")
    assert max(e) < TOL


@pytest.mark.parametrize("B,S,C", [(3, 64, 3), (2, 37, 3), (1, 16, 1)])
def test_obs_uint8_to_float_is_bit_identical_to_the_reference_conversion(L, B, S, C):
    """utils/datasets.py:17: torch.Tensor(obss[i]).permute(2, 0, 1) / 255.0 — here on the GPU from the uploaded uint8 HWC batch"""
    from ocrl_amd.utils.tools import obs_from_uint8
    u8 = torch.randint(0, 256, (B, S, S, C), dtype=torch.uint8, generator=torch.Generator().manual_seed(S))
    ref = torch.stack([torch.Tensor(u8[i].numpy()).permute(2, 0, 1) / 255.0 for i in range(B)])
    got = obs_from_uint8(u8.cuda())
    assert got.shape == (B, C, S, S) and torch.equal(got.cpu(), ref)


@pytest.mark.parametrize("M,N", [(16384, 192), (16416, 576), (32768, 768)])
def test_gemm_model_width_many_rows(L, M, N):
    """K = 192, N a multiple of 192, >= 16k rows (the decoder's products at benchmark scale, several waves of workgroups): NT with
    bias + ReLU and alpha + residual, NN (dX form) with an activation mask"""
    K = 192
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g)
    out = run_gemm(L, dev(A), dev(W), M, N, K, 1, 1, bias=dev(b), relu=1)
    e1 = relerr(out, torch.relu(A.double() @ W.double().T + b.double()))
    out2 = run_gemm(L, dev(A), dev(W), M, N, K, 1, 1, alpha=0.5, resid=dev(R))
    e2 = relerr(out2, 0.5 * (A.double() @ W.double().T) + R.double())
    Wn = torch.randn(K, N, generator=g) / K ** 0.5          # dX = dY W, W stored [N_out = K, K_in = N]
    act = torch.randn(M, N, generator=g)
    out3 = run_gemm(L, dev(A), dev(Wn), M, N, K, 1, 0, mask=dev(act))
    e3 = relerr(out3, (A.double() @ Wn.double()) * (act > 0))
    log(f"gemm model-width {M}x{N}x192: NT bias+relu {e1:.2e} alpha+resid {e2:.2e} NN mask {e3:.2e}")
    assert max(e1, e2, e3) < TOL
