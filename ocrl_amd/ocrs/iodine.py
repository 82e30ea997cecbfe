"""IODINE behind the reference's encoder API (ocrs/iodine/iodine.py:4-14, ocrs/iodine/iodine_module.py:14-271).

``Iodine_Module`` is an ``nn.Module`` container with the reference's parameter names / order (so checkpoints and
``torch.optim.Adam`` state move both ways); the arithmetic of ``_forward`` / ``backward`` / clip + Adam is in
libocrl_hip.so (ocrl_iodine_*).  There is no CPU path."""
import math
from types import SimpleNamespace

import numpy as np
import torch
from torch import nn

from ..dist_utils import allreduce_grads_
from ..engine import IodineEngine
from .base import Base
from .slate import FusedAdam, _Holder


def _dims(ocr_config, env_config):
    c = ocr_config
    fixed = dict(ref_cnn_hidden_size=64, ref_cnn_layers=4, ref_cnn_kernel_size=3, ref_cnn_stride_size=2, dec_cnn_hidden_size=64,
                 dec_cnn_layers=4, dec_cnn_kernel_size=3)
    for k, v in fixed.items():
        if hasattr(c, k) and int(getattr(c, k)) != v:
            raise NotImplementedError(f"ocr.{k}={getattr(c, k)}: the HIP backend implements the reference configuration ({k}={v})")
    if int(getattr(c, "img_channels", 3)) != 3 or int(env_config.obs_channels) != 3:
        raise NotImplementedError("IODINE HIP backend: 3 image channels only")
    return SimpleNamespace(obs_size=int(env_config.obs_size), obs_channels=3, slot_size=int(c.slot_size), num_iterations=int(c.num_iterations),
                           num_slots=int(c.num_slots), sigma=float(c.sigma), beta=float(c.beta), layer_norm=bool(c.layer_norm),
                           ref_mlp_hidden=int(c.ref_mlp_hidden_size))


def _reference_init_(name, t):
    """torch defaults of the reference's layers (nn.Conv2d / nn.Linear / nn.LSTMCell; zero init vectors, iodine_module.py:63-78)"""
    if name in ("slot_mean_init", "slot_logsig_init", "slot_init"):
        return t.zero_()
    if "lstm" in name:
        k = 1.0 / math.sqrt(t.shape[0] // 4)
        return nn.init.uniform_(t, -k, k)
    fan_in = int(np.prod(t.shape[1:])) if t.dim() > 1 else None
    if fan_in is not None:
        return nn.init.kaiming_uniform_(t, a=math.sqrt(5))
    return t          # biases are filled once the matching weight's fan-in is known (see Iodine_Module.__init__)


class Iodine_Module(nn.Module):
    def __init__(self, ocr_config, env_config):
        super().__init__()
        self._dims = _dims(ocr_config, env_config)
        self.slot_size = self._dims.slot_size
        self.num_iterations = self._dims.num_iterations
        self.num_slots = self._dims.num_slots
        self.img_size = self._dims.obs_size
        self.beta, self.sigma = self._dims.beta, self._dims.sigma
        self.use_layernorm = self._dims.layer_norm
        self.rep_dim = self.slot_size
        from .. import _lib
        self._spec = self._query_spec(_lib)
        fan = {}
        for p in self._spec:
            t = _reference_init_(p.name, torch.empty(p.shape))
            if p.name.endswith("weight"):
                fan[p.name[:-len("weight")]] = int(np.prod(p.shape[1:]))
            if p.name.endswith(".bias") and p.name[:-len("bias")] in fan:
                k = 1.0 / math.sqrt(fan[p.name[:-len("bias")]])
                nn.init.uniform_(t, -k, k)
            self._register(p.name, nn.Parameter(t))
        self.engine = None
        self._max_batch = 0
        self._seed = 0
        self._step_seed = 0
        self._injected_noise = None

    def _query_spec(self, _lib):
        import ctypes
        L = _lib.lib()
        d = self._dims
        c = _lib.IodineConfig(d.obs_size, d.obs_channels, d.slot_size, d.num_iterations, d.num_slots, d.sigma, d.beta, int(d.layer_norm),
                              d.ref_mlp_hidden, 1)
        h = ctypes.c_void_p()
        _lib.check(L.ocrl_iodine_create(ctypes.byref(c), ctypes.byref(h)))
        out = []
        name = ctypes.create_string_buffer(256)
        shape = (ctypes.c_int * 4)()
        nd, off, ne = ctypes.c_int(), ctypes.c_longlong(), ctypes.c_longlong()
        for i in range(L.ocrl_iodine_param_count(h)):
            _lib.check(L.ocrl_iodine_param_info(h, i, name, 256, ctypes.byref(shape), ctypes.byref(nd), ctypes.byref(off), ctypes.byref(ne)))
            out.append(SimpleNamespace(name=name.value.decode(), shape=tuple(shape[k] for k in range(nd.value))))
        L.ocrl_iodine_destroy(h)
        return out

    def _get(self, path):
        node = self
        for part in path.split("."):
            if part not in node._modules:
                node.add_module(part, _Holder())
            node = node._modules[part]
        return node

    def _register(self, name, param):
        path, leaf = name.rsplit(".", 1) if "." in name else ("", name)
        (self._get(path) if path else self).register_parameter(leaf, param)

    # ---- device placement: parameters become views of the library's flat buffer
    def to(self, device):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError(f"ocrl_amd Iodine runs on an AMD GPU only (got device={device!r}); there is no CPU path")
        self._device = dev
        self._ensure_engine(max(self._max_batch, 1))
        return self

    def _ensure_engine(self, batch):
        if self.engine is not None and batch <= self._max_batch:
            return
        old = self.engine
        eng = IodineEngine(self._dims, max_batch=batch, device=self._device)
        named = dict(self.named_parameters())
        for p in eng.params:
            eng.view(eng.flat_p, p).copy_(named[p.name].data.to(eng.device))
        if old is not None:
            eng.flat_m.copy_(old.flat_m)
            eng.flat_v.copy_(old.flat_v)
            eng.adam_step = old.adam_step
        for p in eng.params:
            named[p.name].data = eng.view(eng.flat_p, p)
            named[p.name].grad = eng.view(eng.flat_g, p) if p.name != "slot_init" else None     # never receives a gradient (iodine_module.py:76-78)
        self.engine = eng
        self._max_batch = batch
        pending, self._pending_opt = getattr(self, "_pending_opt", None), None
        if pending is not None:       # optimiser state loaded before .to(device)
            pending[0].load_state_dict(pending[1])
        torch.cuda.synchronize(eng.device)

    def _need(self, obs):
        if getattr(self, "_device", None) is None:
            raise RuntimeError("call .to('cuda:N') before using the HIP backend")
        if not obs.is_cuda:
            raise RuntimeError("ocrl_amd: observations must live on the GPU (to_device(batch, device))")
        self._ensure_engine(obs.shape[0])
        return obs.contiguous().float()

    def set_seed(self, seed: int) -> None:
        self._seed = int(seed)
        self._step_seed = 0

    def inject_noise(self, eps):
        """parity hook: [I,B,K,L] N(0,1) draws consumed by the next _forward (the reference's per-iteration rsample)"""
        self._injected_noise = eps

    def _next_seed(self):
        self._step_seed += 1
        return (self._seed << 32) + self._step_seed

    # ---- reference surface
    def _forward(self, image):
        """iodine_module.py:79-252: (slots, recon, recons_masked, masks, loss, mse, kl, means, masks)"""
        image = self._need(image)
        B, K, S, L = image.shape[0], self.num_slots, self.img_size, self.slot_size
        noise, self._injected_noise = self._injected_noise, None
        m = self.engine.forward(image, self._next_seed(), noise).clone()     # fresh metric tensors (the buffer is reused by the next forward)
        eng = self.engine
        slots = eng.tensor("slots", (B, K, L))
        masks = eng.tensor("masks", (B, K, 1, S, S))
        recon = eng.tensor("recon", (B, 3, S, S))
        rmasked = eng.tensor("recons_masked", (B, K, 3, S, S))
        means = eng.tensor("out4", (B, K, S, S, 4))[..., :3].permute(0, 1, 4, 2, 3).clamp(0.0, 1.0)
        return slots, recon, rmasked, masks, m[0], m[1], m[2], means, masks

    def forward(self, obs, with_masks=False):
        slots, _, _, masks, *_ = self._forward(obs)
        if with_masks:
            return slots.clone(), masks.clone()
        return slots.clone()

    def get_loss(self, obs, masks, with_rep=False) -> dict:
        """iodine_module.py:261-269 (masks are required, as in the reference)"""
        _, _, _, attns, loss, mse, kl, _, _ = self._forward(obs)
        fg_mask = 1 - masks[:, -1].unsqueeze(1)
        attns = torch.cat([attns * fg_mask, fg_mask], dim=1)
        from ..utils.tools import calculate_ari
        ari = float(np.mean(calculate_ari(masks, attns)))
        return {"loss": loss, "mse": mse.detach(), "ari": ari, "kld": kl.detach()}

    def backward(self):
        self.engine.backward()

    def get_samples(self, obs) -> dict:
        from ..utils.tools import for_viz, visualize
        slots, recon, recons_masked, masks, loss, mse, kl, means, masks = self._forward(obs)
        return {"samples": for_viz(visualize([obs, recon, recons_masked, masks.repeat(1, 1, 3, 1, 1), means]))}


class _IodineAdam(FusedAdam):
    def step(self, clip=0.0, grad_scale=1.0):
        self._module.engine.clip_adam(self.param_groups[0]["lr"], clip, grad_scale)


class Iodine(Base):
    def __init__(self, ocr_config, env_config) -> None:
        self._module = Iodine_Module(ocr_config, env_config)
        super().__init__(ocr_config, env_config)
        if hasattr(self._config, "learning") and hasattr(self._config.learning, "lr"):      # ocrs/base.py:20-25
            self._opt = _IodineAdam(self._module, [{"params": list(self._module.parameters()), "lr": self._config.learning.lr}])

    def __call__(self, obs, with_masks=False):
        return self._module(obs, with_masks)

    def get_loss(self, obs, masks, with_rep=False) -> dict:
        return self._module.get_loss(obs, masks, False)

    def update(self, obs, masks, step: int) -> dict:
        """ocrs/base.py:60-74: loss -> backward -> [gradient all-reduce] -> clip_grad_norm_(clip, clip_norm_type) -> Adam"""
        if not hasattr(self, "_opt"):
            return {}
        lr = self._config.learning
        # ocrs/base.py:66: a missing clip_norm_type means 'inf' in the reference; this backend implements the L2 clip of configs/ocr/iodine*.yaml only
        if hasattr(lr, "clip") and str(getattr(lr, "clip_norm_type", "inf")) not in ("2", "2.0"):
            raise NotImplementedError("IODINE HIP backend: learning.clip_norm_type must be 2.0 (the reference configuration; absent = 'inf' in ocrs/base.py:66)")
        self._module._step_seed = int(step) * 16      # RNG streams follow the global step, so resume does not replay them
        metrics = self.get_loss(obs, masks)
        self._module.backward()
        scale = allreduce_grads_(self._module.engine.flat_g)
        self._opt.step(lr.clip if hasattr(lr, "clip") else 0.0, scale)
        if hasattr(lr, "clip"):
            metrics["norm"] = self._module.engine.metrics[3] * scale
        return metrics

    def to(self, device) -> None:
        self._module.to(device)
