"""The north-star kernel on its own: ocrl_slot_attention_fwd/bwd (C ABI) against the oracle's slot_attention
(ocrs/common/slot_attn.py:47-102 restated in oracle/slate_oracle.py:294-333) with torch autograd for the gradients.
Ragged N (not a multiple of the 16-position tile), 1..16 slots, slot / MLP widths 64..256; 1..4 attention heads (slot_attn.py:54-92:
the soft-max over heads * slots columns, attn summed over the heads) through ocrl_slot_attention_mh_fwd/bwd."""
import ctypes

import pytest
import torch

from oracle import slate_oracle as O
from tests.gpu_util import log, relerr

pytestmark = pytest.mark.gpu

NAMES = ["norm_inputs.weight", "norm_inputs.bias", "norm_slots.weight", "norm_slots.bias", "norm_mlp.weight", "norm_mlp.bias",
         "project_q.weight", "project_k.weight", "project_v.weight", "gru.weight_ih", "gru.weight_hh", "gru.bias_ih", "gru.bias_hh",
         "mlp.0.weight", "mlp.0.bias", "mlp.2.weight", "mlp.2.bias"]


def shapes(D, H):
    C = 64
    return [(C,), (C,), (D,), (D,), (D,), (D,), (D, D), (D, C), (D, C), (3 * D, D), (3 * D, D), (3 * D,), (3 * D,), (H, D), (H,), (D, H), (D,)]


# the last four shapes take the split forward (several workgroups per image, one launch per iteration): 4, 8, 16 and 8 workgroups per image
@pytest.mark.parametrize("B,N,K,D,H,I,NH", [(3, 200, 5, 128, 192, 3, 1), (2, 1024, 6, 192, 192, 3, 1), (2, 77, 16, 64, 64, 2, 1), (1, 16, 1, 256, 256, 1, 1),
                                            (2, 300, 11, 192, 128, 2, 1), (3, 2100, 6, 192, 192, 3, 1), (2, 4099, 16, 64, 64, 2, 1), (2, 8200, 11, 192, 128, 2, 1),
                                            (1, 4096, 1, 128, 64, 3, 1),
                                            # several heads: 12, 12, 16, 16 and 15 soft-max columns; head widths 96, 32, 16, 96, 64
                                            (2, 1024, 6, 192, 192, 3, 2), (3, 200, 3, 128, 192, 3, 4), (2, 4099, 4, 64, 64, 2, 4), (1, 2100, 8, 192, 128, 2, 2),
                                            (2, 300, 5, 192, 192, 3, 3)])
def test_slot_attention_unit(B, N, K, D, H, I, NH):
    from ocrl_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(B * 1000 + N + K)
    pre = "_slotattn.slot_attention."
    P = {}
    for n, shp in zip(NAMES, shapes(D, H)):
        if n.endswith("weight") and len(shp) == 1:
            P[pre + n] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif len(shp) == 1:
            P[pre + n] = 0.1 * torch.randn(shp, generator=g)
        else:
            P[pre + n] = torch.randn(shp, generator=g) / shp[1] ** 0.5
    x = torch.randn(B, N, 64, generator=g)
    s0 = torch.randn(B, K, D, generator=g)
    dsl = torch.randn(B, K, D, generator=g)
    # ---- reference (CPU, autograd)
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    xr, sr = x.clone().requires_grad_(True), s0.clone().requires_grad_(True)
    slots_ref, attn_ref = O.slot_attention(Pr, xr, sr, I, heads=NH)
    (slots_ref * dsl).sum().backward()
    # ---- device
    dev = lambda t: t.contiguous().cuda()
    wd = [dev(P[pre + n]) for n in NAMES]
    gd = [torch.zeros_like(t) for t in wd]
    xd, s0d, dsd = dev(x), dev(s0), dev(dsl)
    slots = torch.empty(B, K, D, device="cuda"); attn = torch.empty(B, N, K, device="cuda")
    dx = torch.empty(B, N, 64, device="cuda"); ds0 = torch.empty(B, K, D, device="cuda")
    nws = L.ocrl_slot_attention_mh_ws_floats(B, N, K, D, H, I, NH)
    if NH == 1:
        assert nws == L.ocrl_slot_attention_ws_floats(B, K, D, H, I)
    ws = torch.empty(nws, device="cuda")
    arr = (ctypes.c_void_p * 17)(*[t.data_ptr() for t in wd])
    garr = (ctypes.c_void_p * 17)(*[t.data_ptr() for t in gd])
    p = _lib.ptr
    if NH == 1:         # the single-head entry points
        _lib.check(L.ocrl_slot_attention_fwd(p(xd), p(s0d), arr, p(slots), p(attn), B, N, K, D, H, I, p(ws), nws, None))
        _lib.check(L.ocrl_slot_attention_bwd(p(xd), p(dsd), p(dx), p(ds0), garr, B, N, K, D, H, I, p(ws), nws, None))
    else:
        _lib.check(L.ocrl_slot_attention_mh_fwd(p(xd), p(s0d), arr, p(slots), p(attn), B, N, K, D, H, I, NH, p(ws), nws, None))
        _lib.check(L.ocrl_slot_attention_mh_bwd(p(xd), p(dsd), p(dx), p(ds0), garr, B, N, K, D, H, I, NH, p(ws), nws, None))
    torch.cuda.synchronize()
    e = dict(slots=relerr(slots, slots_ref), attn=relerr(attn, attn_ref.reshape(B, N, K)), dx=relerr(dx, xr.grad), dslots0=relerr(ds0, sr.grad))
    gmax = max(float(Pr[pre + n].grad.abs().max()) for n in NAMES)
    # norm_slots.bias has an exactly zero gradient (a common shift of all queries cancels in the softmax over slots): both sides
    # hold rounding noise there, hence the floor relative to the largest gradient
    ge = {n: relerr(t, Pr[pre + n].grad, floor=1e-4 * gmax) for n, t in zip(NAMES, gd)}
    worst = max(ge, key=ge.get)
    log(f"[slot_attention unit B{B} N{N} K{K} D{D} H{H} I{I} heads{NH}] " + " ".join(f"{k}={v:.2e}" for k, v in e.items()) + f" worst dW {worst}={ge[worst]:.2e}")
    assert e["slots"] < 1e-4 and e["attn"] < 1e-4 and e["dx"] < 1e-3 and e["dslots0"] < 1e-3
    assert ge[worst] < 1e-3, ge
