"""The arithmetic behind the exploratory split-precision convolutions (ocrl_amd/csrc/conv_x3.hip), checked in numpy: every fp32 number is
the EXACT sum of three bf16 numbers (h = bf16(x) rounded to nearest, m = bf16(x - h), l = x - h - m), and a product accumulated from the
six kept plane products h*h' + h*m' + m*h' + m*m' + h*l' + l*h' differs from the exact product by less than 2^-23 of |x*w| (the dropped
m*l' and l*m' are below 2^-24 each: |m| <= 2^-8 |x|, |l| <= 2^-16 |x|) -- the order of one fp32 rounding.  (The kernels do the same bit operations in registers; their results against fp64 are in tests/test_gpu_kernels.py.)"""
import numpy as np


def bf16_rne(x):
    """float32 -> the nearest bf16 (ties to even), returned as float32"""
    u = x.view(np.uint32).astype(np.uint64)
    u = (u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))) & np.uint64(0xFFFF0000)
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    """x (float32) -> three float32 arrays, each exactly representable in bf16 (low 16 bits zero), x = h + m + l:
    h = bf16(x), m = bf16(x - h), l = x - h - m, as csrc/conv_x3.hip::x3_split does with v_cvt_pk_bf16_f32"""
    h = bf16_rne(x)
    r = x - h                                       # exact
    m = bf16_rne(r)
    l = r - m                                       # exact, at most 8 significant bits
    return h, m, l


def test_three_bf16_planes_reconstruct_fp32_exactly():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(1 << 20) * np.exp(4.0 * rng.standard_normal(1 << 20))).astype(np.float32)
    x[:8] = [0.0, -0.0, 1.0, -1.0, 3.0e-30, -7.5e20, np.float32(1) + np.float32(2) ** -23, np.float32(2) ** -100]
    h, m, l = split3(x)
    for p in (h, m, l):
        assert not (p.view(np.uint32) & np.uint32(0xFFFF)).any()          # each plane is a bf16 number
    assert np.array_equal((h.astype(np.float64) + m.astype(np.float64)) + l.astype(np.float64), x.astype(np.float64))
    nz = x != 0
    assert (np.abs(m[nz]) <= np.abs(x[nz]) * 2.0 ** -8).all() and (np.abs(l[nz]) <= np.abs(x[nz]) * 2.0 ** -16).all()


def test_six_plane_products_are_fp32_equivalent():
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(1 << 18) * np.exp(2.0 * rng.standard_normal(1 << 18))).astype(np.float32)
    w = (rng.standard_normal(1 << 18) / 40.0).astype(np.float32)
    xh, xm, xl = (p.astype(np.float64) for p in split3(x))
    wh, wm, wl = (p.astype(np.float64) for p in split3(w))
    kept = xh * wh + xh * wm + xm * wh + xm * wm + xh * wl + xl * wh
    exact = x.astype(np.float64) * w.astype(np.float64)
    rel = np.abs(kept - exact) / np.abs(exact)
    assert rel.max() < 2.0 ** -23, rel.max()                               # dropped: m*l' + l*m' + l*l' <= 2 * 2^-8 * 2^-16 |x w| (+ 2^-32)
    assert np.median(rel) < 2.0 ** -24                                     # typical products are far inside the bound
    # and a dot product of 1600 terms (one 5x5 x 64-channel output) accumulated in fp32 stays at fp32 accuracy
    X, W = x[:160000].reshape(100, 1600), w[:160000].reshape(100, 1600)
    planes = [p.reshape(100, 1600) for p in split3(X.ravel())], [p.reshape(100, 1600) for p in split3(W.ravel())]
    acc = np.zeros(100, np.float32)
    for a, b in ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)):          # the kernels' order: small terms first
        acc += np.einsum("ij,ij->i", planes[0][a], planes[1][b], dtype=np.float32)
    ref = np.einsum("ij,ij->i", X.astype(np.float64), W.astype(np.float64))
    fp32 = np.einsum("ij,ij->i", X, W, dtype=np.float32)
    scale = np.abs(X.astype(np.float64) * W.astype(np.float64)).sum(1)
    assert (np.abs(acc - ref) / scale).max() < 4 * max((np.abs(fp32 - ref) / scale).max(), 2.0 ** -24)
