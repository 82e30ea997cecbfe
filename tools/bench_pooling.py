"""Latency of the slot-set pooling head through the C ABI (ocrl_pool_transformer_fwd/_bwd): default reference config
(6 slots x 192 -> d_model 128, 8 heads, ff 2048, 1 layer).  B = rollout batches (num_envs) and a PPO minibatch."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrl_amd import _lib
from types import SimpleNamespace
from ocrl_amd.poolings.transformer import Transformer_Module
L = _lib.lib()
p = _lib.ptr
cfg = SimpleNamespace(rep_dim=192, num_slots=6, d_model=128, nhead=8, num_layers=1, dim_feedforward=2048, dropout=0.1, pos_emb="None")
torch.manual_seed(0)
w = [t.detach().cuda().contiguous() for t in Transformer_Module(cfg.rep_dim, cfg.num_slots, cfg)._param_list()]; g = [torch.empty_like(t) for t in w]
arr = (ctypes.c_void_p * len(w))(*[t.data_ptr() for t in w]); garr = (ctypes.c_void_p * len(g))(*[t.data_ptr() for t in g])
K, Din, d, h, ff, nl = cfg.num_slots, cfg.rep_dim, cfg.d_model, cfg.nhead, cfg.dim_feedforward, cfg.num_layers
def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B in (4, 16, 32, 256, 2048):
    x = torch.randn(B, K, Din, device="cuda"); out = torch.empty(B, d, device="cuda"); dc = torch.randn(B, d, device="cuda"); dx = torch.empty_like(x)
    n = L.ocrl_pool_transformer_ws_floats(B, K, d, h, ff, nl); ws = torch.empty(n, device="cuda")
    fwd = lambda: _lib.check(L.ocrl_pool_transformer_fwd(p(x), arr, None, p(out), B, K, Din, d, h, ff, nl, 0.0, 0, p(ws), n, None))
    trn = lambda: (_lib.check(L.ocrl_pool_transformer_fwd(p(x), arr, None, p(out), B, K, Din, d, h, ff, nl, 0.1, 7, p(ws), n, None)),
                   _lib.check(L.ocrl_pool_transformer_bwd(p(x), p(dc), arr, None, garr, B, K, Din, d, h, ff, nl, 0.1, 7, p(ws), n, None)))
    print(f"pooling B{B}: eval forward {t(fwd):.1f} us, train forward+backward {t(trn):.1f} us")
