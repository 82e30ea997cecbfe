"""SlateEngine — owns the C handle, the flat parameter / gradient / Adam buffers (torch-allocated,
adopted by the library) and the workspace.  Host-side plumbing only; all arithmetic is in HIP."""
import ctypes
from types import SimpleNamespace

import torch

from . import _lib

SITE_ZPOS = 1
SITE_BLK_BASE = 16


def _aligned_empty(nbytes, device):
    """uint8 tensor whose data_ptr is 256-byte aligned"""
    raw = torch.empty(nbytes + 512, dtype=torch.uint8, device=device)
    off = (-raw.data_ptr()) % 256
    return raw[off:off + nbytes]


class SlateEngine:
    def __init__(self, dims, max_batch, device="cuda:0", with_optimizer=True):
        """dims: namespace with obs_size, obs_channels, vocab_size, d_model, cnn_hidden, num_slots,
        num_iterations, slot_size, mlp_hidden, num_dec_blocks, num_dec_heads, dropout."""
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError(f"ocrl_amd runs on an AMD GPU only (device={device!r}); there is no CPU path")
        if not torch.cuda.is_available():
            raise RuntimeError("ocrl_amd: no GPU visible to PyTorch-ROCm")
        self.L = _lib.lib()
        self.device = dev
        self.dims = dims
        self.max_batch = int(max_batch)
        c = _lib.SlateConfig(dims.obs_size, dims.obs_channels, dims.vocab_size, dims.d_model, dims.cnn_hidden, dims.num_slots,
                             dims.num_iterations, dims.slot_size, dims.mlp_hidden, dims.num_dec_blocks, dims.num_dec_heads,
                             float(dims.dropout), self.max_batch, int(bool(getattr(dims, "use_bcdec", False))),
                             int(bool(getattr(dims, "hard", False))), int(getattr(dims, "num_slot_heads", 1)))
        h = ctypes.c_void_p()
        _lib.check(self.L.ocrl_slate_create(ctypes.byref(c), ctypes.byref(h)))
        self.h = h
        self.params = []          # list of SimpleNamespace(name, shape, offset, numel, group)
        name = ctypes.create_string_buffer(256)
        shape = (ctypes.c_int * 4)()
        nd, off, ne, grp = ctypes.c_int(), ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_int()
        for i in range(self.L.ocrl_slate_param_count(h)):
            _lib.check(self.L.ocrl_slate_param_info(h, i, name, 256, ctypes.byref(shape), ctypes.byref(nd), ctypes.byref(off),
                                                    ctypes.byref(ne), ctypes.byref(grp)))
            self.params.append(SimpleNamespace(name=name.value.decode(), shape=tuple(shape[k] for k in range(nd.value)),
                                               offset=off.value, numel=ne.value, group=grp.value))
        self.flat_size = self.L.ocrl_slate_flat_size(h)
        self.group_begin = [self.L.ocrl_slate_group_begin(h, g) for g in range(4)]
        with torch.cuda.device(dev):
            mk = lambda: _aligned_empty(self.flat_size * 4, dev).view(torch.float32).zero_()
            self.flat_p, self.flat_g = mk(), mk()
            self.flat_m, self.flat_v = (mk(), mk()) if with_optimizer else (None, None)
            self.ws_bytes = self.L.ocrl_slate_workspace_bytes(h)
            self.ws = _aligned_empty(self.ws_bytes, dev)
            _lib.check(self.L.ocrl_slate_bind(h, _lib.ptr(self.flat_p), _lib.ptr(self.flat_g), _lib.ptr(self.flat_m), _lib.ptr(self.flat_v),
                                              _lib.ptr(self.ws), self.ws_bytes))
        mp = self.L.ocrl_slate_metrics(h)
        self.metrics = self._view(mp, 8, torch.float32)
        self.adam_step = 0

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.ocrl_slate_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- views
    def _view(self, p, count, dtype):
        off = int(p) - self.ws.data_ptr()
        assert 0 <= off and off + count * 4 <= self.ws.numel(), "pointer outside the workspace"
        return self.ws[off:off + count * 4].view(dtype)

    def view(self, flat, p):
        return flat[p.offset:p.offset + p.numel].view(p.shape)

    def param(self, name):
        p = next(q for q in self.params if q.name == name)
        return self.view(self.flat_p, p)

    def grad(self, name):
        p = next(q for q in self.params if q.name == name)
        return self.view(self.flat_g, p)

    def tensor(self, name, shape, dtype=torch.float32):
        if name == "z":      # the soft sample is not a by-product of the step (fused soft-max heads): written on request
            _lib.check(self.L.ocrl_slate_soft_z(self.h, self.stream))
        p, n = ctypes.c_void_p(), ctypes.c_longlong()
        _lib.check(self.L.ocrl_slate_tensor(self.h, name.encode(), ctypes.byref(p), ctypes.byref(n)))
        cnt = 1
        for s in shape:
            cnt *= s
        assert cnt <= n.value, (name, shape, n.value)
        return self._view(p.value, cnt, dtype).view(shape)

    @property
    def stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- step pieces
    def forward(self, obs, tau, train, seed, noise=None):
        """obs [B,3,S,S] fp32 contiguous on device.  noise: dict(z=[B,T,V], z_hard=[B,T,V], slots=[B,K,D]) or None."""
        assert obs.is_cuda and obs.dtype == torch.float32 and obs.is_contiguous()
        nz = nzh = ns = None
        if noise is not None:
            nz, nzh, ns = noise.get("z"), noise.get("z_hard"), noise.get("slots")
            for t in (nz, nzh, ns):
                assert t is None or (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous())
        self._keep = (obs, nz, nzh, ns)      # keep inputs alive until the stream has consumed them
        _lib.check(self.L.ocrl_slate_forward(self.h, _lib.ptr(obs), obs.shape[0], float(tau), int(bool(train)), int(seed),
                                             _lib.ptr(nz), _lib.ptr(nzh), _lib.ptr(ns), self.stream))
        return self.metrics

    def backward(self):
        _lib.check(self.L.ocrl_slate_backward(self.h, self.stream))

    def generate(self):
        _lib.check(self.L.ocrl_slate_generate(self.h, self.stream))

    def encode(self, obs, seed, slot_noise=None):
        assert obs.is_cuda and obs.dtype == torch.float32 and obs.is_contiguous()
        self._keep = (obs, slot_noise)
        self.encode_generation = getattr(self, "encode_generation", 0) + 1      # which encode() the saved activations belong to
        _lib.check(self.L.ocrl_slate_encode(self.h, _lib.ptr(obs), obs.shape[0], int(seed), _lib.ptr(slot_noise), self.stream))

    def freeze_weights(self, on=True):
        """the parameters will not change: encode() keeps its derived weight images between calls"""
        _lib.check(self.L.ocrl_slate_freeze_weights(self.h, int(bool(on))))

    def encode_backward(self, dslots):
        """d loss / d slots of the last encode() -> flat_g (encoder tensors; zeros elsewhere)"""
        assert dslots.is_cuda and dslots.dtype == torch.float32 and dslots.is_contiguous()
        _lib.check(self.L.ocrl_slate_encode_backward(self.h, _lib.ptr(dslots), self.stream))

    def clip_adam(self, lrs, clip, grad_scale=1.0):
        self.adam_step += 1
        arr = (ctypes.c_float * 3)(*[float(x) for x in lrs])
        _lib.check(self.L.ocrl_slate_clip_adam(self.h, ctypes.byref(arr), float(clip if clip is not None else 0.0), self.adam_step,
                                               float(grad_scale), self.stream))

    def grad_norm(self):
        _lib.check(self.L.ocrl_slate_grad_norm(self.h, self.stream))
        return self.metrics[3]

    def dropout_mask(self, site, shape):
        n = 1
        for s in shape:
            n *= s
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        _lib.check(self.L.ocrl_slate_dropout_mask(self.h, int(site), n, _lib.ptr(out), self.stream))
        return out.view(shape)


class IodineEngine:
    """Same role as SlateEngine for the IODINE handle (include/ocrl_hip.h: ocrl_iodine_*)."""

    def __init__(self, dims, max_batch, device="cuda:0", with_optimizer=True):
        """dims: namespace with obs_size, obs_channels, slot_size, num_iterations, num_slots, sigma, beta, layer_norm, ref_mlp_hidden."""
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError(f"ocrl_amd runs on an AMD GPU only (device={device!r}); there is no CPU path")
        if not torch.cuda.is_available():
            raise RuntimeError("ocrl_amd: no GPU visible to PyTorch-ROCm")
        self.L = _lib.lib()
        self.device = dev
        self.dims = dims
        self.max_batch = int(max_batch)
        c = _lib.IodineConfig(dims.obs_size, dims.obs_channels, dims.slot_size, dims.num_iterations, dims.num_slots, float(dims.sigma),
                              float(dims.beta), int(bool(dims.layer_norm)), dims.ref_mlp_hidden, self.max_batch)
        h = ctypes.c_void_p()
        _lib.check(self.L.ocrl_iodine_create(ctypes.byref(c), ctypes.byref(h)))
        self.h = h
        self.params = []
        name = ctypes.create_string_buffer(256)
        shape = (ctypes.c_int * 4)()
        nd, off, ne = ctypes.c_int(), ctypes.c_longlong(), ctypes.c_longlong()
        for i in range(self.L.ocrl_iodine_param_count(h)):
            _lib.check(self.L.ocrl_iodine_param_info(h, i, name, 256, ctypes.byref(shape), ctypes.byref(nd), ctypes.byref(off), ctypes.byref(ne)))
            self.params.append(SimpleNamespace(name=name.value.decode(), shape=tuple(shape[k] for k in range(nd.value)), offset=off.value,
                                               numel=ne.value, group=0))
        self.flat_size = self.L.ocrl_iodine_flat_size(h)
        with torch.cuda.device(dev):
            mk = lambda: _aligned_empty(self.flat_size * 4, dev).view(torch.float32).zero_()
            self.flat_p, self.flat_g = mk(), mk()
            self.flat_m, self.flat_v = (mk(), mk()) if with_optimizer else (None, None)
            self.ws_bytes = self.L.ocrl_iodine_workspace_bytes(h)
            self.ws = _aligned_empty(self.ws_bytes, dev)
            _lib.check(self.L.ocrl_iodine_bind(h, _lib.ptr(self.flat_p), _lib.ptr(self.flat_g), _lib.ptr(self.flat_m), _lib.ptr(self.flat_v),
                                               _lib.ptr(self.ws), self.ws_bytes))
        self.metrics = self._view(self.L.ocrl_iodine_metrics(h), 8, torch.float32)
        self.adam_step = 0

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.ocrl_iodine_destroy(self.h)
                self.h = None
        except Exception:
            pass

    _view = SlateEngine._view
    view = SlateEngine.view
    param = SlateEngine.param
    grad = SlateEngine.grad
    stream = SlateEngine.stream

    def tensor(self, name, shape, dtype=torch.float32):
        p, n = ctypes.c_void_p(), ctypes.c_longlong()
        _lib.check(self.L.ocrl_iodine_tensor(self.h, name.encode(), ctypes.byref(p), ctypes.byref(n)))
        cnt = 1
        for s in shape:
            cnt *= s
        assert cnt <= n.value, (name, shape, n.value)
        return self._view(p.value, cnt, dtype).view(shape)

    def forward(self, obs, seed, noise=None):
        """obs [B,3,S,S] fp32 contiguous on device; noise: optional [I,B,K,L] N(0,1) draws."""
        assert obs.is_cuda and obs.dtype == torch.float32 and obs.is_contiguous()
        assert noise is None or (noise.is_cuda and noise.dtype == torch.float32 and noise.is_contiguous())
        self._keep = (obs, noise)
        _lib.check(self.L.ocrl_iodine_forward(self.h, _lib.ptr(obs), obs.shape[0], int(seed), _lib.ptr(noise), self.stream))
        return self.metrics

    def backward(self):
        _lib.check(self.L.ocrl_iodine_backward(self.h, self.stream))

    def clip_adam(self, lr, clip, grad_scale=1.0):
        self.adam_step += 1
        _lib.check(self.L.ocrl_iodine_clip_adam(self.h, float(lr), float(clip if clip is not None else 0.0), self.adam_step, float(grad_scale),
                                                self.stream))

    def grad_norm(self):
        _lib.check(self.L.ocrl_iodine_grad_norm(self.h, self.stream))
        return self.metrics[3]
