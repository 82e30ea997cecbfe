"""SLATE / Slot-Attention encoder on the HIP backend, behind the reference's Python surface
(ocrs/slate/slate.py:13-69, ocrs/slate/slate_module.py:23-267).

``SLATE_Module`` is an ``nn.Module`` only as a *container*: its Parameters are views into the flat
device buffer the C library computes on (so ``parameters()``, ``state_dict()`` and checkpoints keep
the reference's names and shapes); forward/backward never run through ATen.
"""
import math
from types import SimpleNamespace

import torch
from torch import nn

from ..dist_utils import allreduce_grads_
from ..engine import SlateEngine
from .base import Base


# schedules: ocrs/common/utils.py:37-65
def cosine_anneal(step, start_value, final_value, start_step, final_step):
    assert start_value >= final_value and start_step <= final_step
    if step < start_step:
        return start_value
    if step >= final_step:
        return final_value
    a = 0.5 * (start_value - final_value)
    b = 0.5 * (start_value + final_value)
    return a * math.cos(math.pi * (step - start_step) / (final_step - start_step)) + b


def linear_warmup(step, start_value, final_value, start_step, final_step):
    assert start_value <= final_value and start_step <= final_step
    if step < start_step:
        return start_value
    if step >= final_step:
        return final_value
    return (final_value - start_value) * (step + 1 - start_step) / (final_step - start_step) + start_value


def _dims(ocr_config, env_config):
    sa = ocr_config.slotattr
    heads = int(getattr(sa, "num_slot_heads", 1))
    if heads != 1:
        # ocrs/common/slot_attn.py:54-92 splits q/k/v into `heads` groups and takes the softmax over heads*slots columns: the HIP
        # kernels run those columns on the 16 columns of one MFMA tile, which bounds the supported shapes
        K, D = int(sa.num_slots), int(sa.slot_size)
        if heads < 1 or K > 8 or heads * K > 16 or D % heads or (D // heads) % 16:
            raise NotImplementedError(f"ocrl_amd: ocr.slotattr.num_slot_heads={heads} with num_slots={K}, slot_size={D} is not supported by the "
                                      "HIP backend: heads * num_slots <= 16, num_slots <= 8 and a head width that is a multiple of 16 are")
    return SimpleNamespace(
        obs_size=int(env_config.obs_size), obs_channels=int(env_config.obs_channels),
        vocab_size=int(ocr_config.dvae.vocab_size), d_model=int(ocr_config.dvae.d_model), cnn_hidden=int(ocr_config.cnn.hidden_size),
        num_slots=int(sa.num_slots), num_iterations=int(sa.num_iterations), slot_size=int(sa.slot_size),
        mlp_hidden=int(sa.mlp_hidden_size), num_dec_blocks=int(ocr_config.tfdec.num_dec_blocks),
        num_dec_heads=int(ocr_config.tfdec.num_dec_heads), dropout=float(ocr_config.learning.dropout),
        use_bcdec=bool(ocr_config.use_bcdec), hard=bool(ocr_config.hard), num_slot_heads=heads)


def _position_grid(S):
    """ocrs/common/utils.py:14-27, channel order [north, south, west, east]"""
    lin = torch.linspace(0, 1, S)
    east = lin.view(1, S).expand(S, S)
    west = torch.linspace(1, 0, S).view(1, S).expand(S, S)
    south = lin.view(S, 1).expand(S, S)
    north = torch.linspace(1, 0, S).view(S, 1).expand(S, S)
    return torch.stack([north, south, west, east], 0).unsqueeze(0).contiguous()


def reference_init_(name, t, num_blocks):
    """Initialisers of the reference layer factories (ocrs/common/networks.py:6-74, slot_attn.py:133-136,
    transformer.py:53-58,193-198, slate_module.py:273,287-288): statistical, not bitwise, parity."""
    init = nn.init
    gain = (3 * num_blocks) ** (-0.5)
    if name.endswith("bias") or name.endswith("bias_ih") or name.endswith("bias_hh"):
        if name == "_enc_pos.channels_map.bias":
            init.uniform_(t, -0.5, 0.5)                     # nn.Conv2d default: U(+-1/sqrt(fan_in)), fan_in = 4
        else:
            init.zeros_(t)
    elif "norm" in name and name.endswith("weight"):
        init.ones_(t)
    elif name == "_enc_pos.channels_map.weight":
        init.kaiming_uniform_(t, a=math.sqrt(5))
    elif name == "_dict.dictionary.weight":
        init.normal_(t)
    elif name == "_z_pos.pe":
        init.trunc_normal_(t.zero_())
    elif name.endswith("gru.weight_hh"):
        init.orthogonal_(t)
    elif name.endswith(".m.weight") or name.endswith("mlp.0.weight") or name.endswith("ffn.0.weight"):
        init.kaiming_uniform_(t, nonlinearity="relu")       # Conv2dBlock / first MLP layers
    elif name.endswith("proj_o.weight") or name.endswith("ffn.2.weight"):
        init.xavier_uniform_(t, gain)
    else:
        init.xavier_uniform_(t)
    return t


class _Holder(nn.Module):
    """plain container node used to reproduce the reference's dotted state_dict names"""


class SLATE_Module(nn.Module):
    def __init__(self, ocr_config, env_config) -> None:
        super().__init__()
        self._obs_size = int(env_config.obs_size)
        self._obs_channels = int(env_config.obs_channels)
        self._use_cnn_feat = bool(ocr_config.use_cnn_feat)
        self._use_bcdec = bool(ocr_config.use_bcdec)
        self._dims = _dims(ocr_config, env_config)
        self._vocab_size = self._dims.vocab_size
        self._num_slots = self._dims.num_slots
        self._enc_size = self._obs_size // 4
        self._tau_start, self._tau_final, self._tau_steps = ocr_config.tau_start, ocr_config.tau_final, ocr_config.tau_steps
        self._tau = 1.0
        self._hard = bool(ocr_config.hard)
        if self._use_cnn_feat:            # slate_module.py:87-92
            self.num_slots = self._obs_size ** 2
            self.rep_dim = self._dims.cnn_hidden + self._obs_channels
        else:
            self.num_slots = self._dims.num_slots
            self.rep_dim = self._dims.slot_size
        # parameter containers in the reference's order / names; CPU tensors until .to(cuda)
        from .. import _lib
        self._spec = self._query_spec(_lib)
        T = self._enc_size ** 2
        self._pnames = []
        block_masks = {}
        for p in self._spec:
            t = reference_init_(p.name, torch.empty(p.shape), self._dims.num_dec_blocks)
            self._register(p.name, nn.Parameter(t))
            self._pnames.append(p.name)
            if p.name.endswith("self_attn_layer_norm.weight"):
                block_masks[p.name.rsplit(".", 2)[0]] = True
        # bool mask Parameters (transformer.py:151-152) and the position buffers keep checkpoints interchangeable
        for blk in block_masks:
            mask = torch.triu(torch.ones((T, T), dtype=torch.bool), diagonal=1)
            self._register(blk + ".self_attn_mask", nn.Parameter(mask, requires_grad=False), first=True)
        self._get("_enc_pos").register_buffer("linear_position_embedding", _position_grid(self._obs_size))
        self.engine = None
        self._max_batch = 0
        self._seed = 0
        self._step_seed = 0
        self._injected_noise = None
        self.finetune_through_slots = False      # set by poolings.Base when learn_downstream_loss=True: forward() returns attached slots

    # ---- container plumbing
    def _query_spec(self, _lib):
        import ctypes
        L = _lib.lib()
        d = self._dims
        c = _lib.SlateConfig(d.obs_size, d.obs_channels, d.vocab_size, d.d_model, d.cnn_hidden, d.num_slots, d.num_iterations,
                             d.slot_size, d.mlp_hidden, d.num_dec_blocks, d.num_dec_heads, d.dropout, 1, int(d.use_bcdec), int(d.hard),
                             int(getattr(d, "num_slot_heads", 1)))
        h = ctypes.c_void_p()
        _lib.check(L.ocrl_slate_create(ctypes.byref(c), ctypes.byref(h)))
        out = []
        name = ctypes.create_string_buffer(256)
        shape = (ctypes.c_int * 4)()
        nd, off, ne, grp = ctypes.c_int(), ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_int()
        for i in range(L.ocrl_slate_param_count(h)):
            _lib.check(L.ocrl_slate_param_info(h, i, name, 256, ctypes.byref(shape), ctypes.byref(nd), ctypes.byref(off), ctypes.byref(ne), ctypes.byref(grp)))
            out.append(SimpleNamespace(name=name.value.decode(), shape=tuple(shape[k] for k in range(nd.value)), group=grp.value))
        L.ocrl_slate_destroy(h)
        return out

    def _get(self, path):
        node = self
        for part in path.split("."):
            if part not in node._modules:
                node.add_module(part, _Holder())
            node = node._modules[part]
        return node

    def _register(self, name, param, first=False):
        path, leaf = name.rsplit(".", 1) if "." in name else ("", name)
        node = self._get(path) if path else self
        node.register_parameter(leaf, param)
        if first:       # the mask is the block's first attribute in the reference -> first in state_dict order
            items = list(node._parameters.items())
            node._parameters.clear()
            node._parameters[leaf] = param
            for k, v in items:
                if k != leaf:
                    node._parameters[k] = v

    def _named_trainable(self):
        d = dict(self.named_parameters())
        return [(n, d[n]) for n in self._pnames]

    def get_dvae_params(self):
        return [p for (n, p), s in zip(self._named_trainable(), self._spec) if s.group == 0]

    def get_sa_params(self):
        return [p for (n, p), s in zip(self._named_trainable(), self._spec) if s.group == 1]

    def get_tfdec_params(self):
        """slate_module.py:115-121: includes the (grad-less) bool mask Parameters, in module order"""
        d = dict(self.named_parameters())
        out = []
        for s in self._spec:
            if s.group != 2:
                continue
            if s.name.endswith("self_attn_layer_norm.weight"):
                out.append(d[s.name.rsplit(".", 2)[0] + ".self_attn_mask"])
            out.append(d[s.name])
        return out

    # ---- device placement: parameters become views of the library's flat buffer
    def to(self, device):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError(f"ocrl_amd SLATE runs on an AMD GPU only (got device={device!r}); there is no CPU path")
        self._device = dev
        self._ensure_engine(max(self._max_batch, 1))
        return self

    def _ensure_engine(self, batch):
        if self.engine is not None and batch <= self._max_batch:
            return
        old = self.engine
        eng = SlateEngine(self._dims, max_batch=batch, device=self._device)
        named = dict(self.named_parameters())
        for p in eng.params:
            eng.view(eng.flat_p, p).copy_(named[p.name].data.to(eng.device))
        if old is not None:
            eng.flat_m.copy_(old.flat_m)
            eng.flat_v.copy_(old.flat_v)
            eng.adam_step = old.adam_step
        for p in eng.params:
            named[p.name].data = eng.view(eng.flat_p, p)
            named[p.name].grad = eng.view(eng.flat_g, p)
        for n, q in self.named_parameters():
            if q.dtype == torch.bool:
                q.data = q.data.to(eng.device)
        for n, b in self.named_buffers():
            b.data = b.data.to(eng.device)
        self.engine = eng
        self._max_batch = batch
        if getattr(self, "_frozen", False):
            eng.freeze_weights(True)
        pending, self._pending_opt = getattr(self, "_pending_opt", None), None
        if pending is not None:       # optimiser state loaded before .to(device), as the reference's callers do (sb3s/ocr_extractor.py:33-36)
            pending[0].load_state_dict(pending[1])
        torch.cuda.synchronize(eng.device)

    def _need(self, obs):
        if getattr(self, "_device", None) is None:
            raise RuntimeError("call .to('cuda:N') before using the HIP backend")
        if not obs.is_cuda:
            raise RuntimeError("ocrl_amd: observations must live on the GPU (to_device(batch, device))")
        self._ensure_engine(obs.shape[0])
        return obs.contiguous().float()

    def freeze_weights(self, on=True):
        """Serving with a frozen, pre-trained encoder (sb3s/ocr_extractor.py:33-36 without finetuning): promise that the parameters do not
        change, so model(obs) stops rebuilding the derived weight images at every call.  Call after the checkpoint is loaded."""
        self._frozen = bool(on)
        if self.engine is not None:
            self.engine.freeze_weights(self._frozen)

    def _publish_encoder_grads(self):
        """after ocrl_slate_encode_backward: the parameters' .grad are views of the flat gradient buffer (set in _ensure_engine); an
        optimiser's zero_grad(set_to_none=True) drops them, so the tensors that just received a gradient get their view back.  The
        others stay None and a torch optimiser skips them, as it does for the reference's unused parameters."""
        eng = self.engine
        named = dict(self.named_parameters())
        for p in eng.params:
            if p.name.startswith(("_enc.", "_enc_pos.", "_slotattn.")):
                t = named[p.name]
                view = eng.view(eng.flat_g, p)
                if t.grad is None:
                    t.grad = view
                elif t.grad.data_ptr() != view.data_ptr():
                    t.grad.add_(view)

    def _grad_hook(self):
        """a one-element leaf that requires grad: gives _EncodeGrad a differentiable input, so autograd calls its backward"""
        h = getattr(self, "_grad_hook_t", None)
        if h is None or h.device != self._device:
            h = self._grad_hook_t = torch.zeros(1, device=self._device, requires_grad=True)
        return h

    # ---- reference surface
    def update_tau(self, step: int) -> None:
        self._tau = cosine_anneal(step, self._tau_start, self._tau_final, 0, self._tau_steps)

    def set_seed(self, seed: int) -> None:
        self._seed = int(seed)
        self._step_seed = 0

    def inject_noise(self, noise):
        """parity hook: dict(z=[B,T,V], z_hard=[B,T,V], slots=[B,K,D]) consumed by the next get_loss/forward"""
        self._injected_noise = noise

    def _next_seed(self):
        """device-RNG stream id of the next forward.  update() sets the counter to 2*step, so a training step draws from the odd id
        2*step + 1; every further forward before the next update() (a validation pass of any length) moves on in steps of 2 from a
        separate high-bit range, so it never lands on a training step's stream"""
        first = (self._step_seed & 1) == 0
        self._step_seed += 1 if first else 2
        if first:
            return (self._seed << 32) + self._step_seed
        return (self._seed << 32) + (1 << 31) + self._step_seed

    def _attns_image(self, B):
        K, S = self._num_slots, self._obs_size
        a = self.engine.tensor("attn", (B, S * S, K))
        return a.transpose(-1, -2).reshape(B, K, 1, S, S)

    def forward(self, obs, with_attns=False, with_masks=False):
        """slate_module.py:181-196"""
        assert not (with_attns and with_masks)
        obs = self._need(obs)
        B = obs.shape[0]
        noise, self._injected_noise = self._injected_noise, None
        self.engine.encode(obs, self._next_seed(), None if noise is None else noise.get("slots"))
        if self._use_cnn_feat:
            feats = self.engine.tensor("feats", (B, self._obs_size ** 2, self._dims.cnn_hidden))
            return torch.cat([feats, obs.permute(0, 2, 3, 1).reshape(B, -1, obs.shape[1])], dim=-1)
        slots = self.engine.tensor("slots", (B, self._num_slots, self._dims.slot_size)).clone()
        if getattr(self, "finetune_through_slots", False) and torch.is_grad_enabled():
            # poolings/base.py:53-55 (learn_downstream_loss): the slots stay attached, so a downstream loss reaches the encoder
            slots = _EncodeGrad.apply(self._grad_hook(), slots, self, self.engine.encode_generation)
        if with_attns or with_masks:
            attns = self._attns_image(B).clone()
            if with_attns:
                attns = obs.unsqueeze(1) * attns + (1.0 - attns)
            return slots, attns
        return slots

    def get_loss(self, obs, masks, with_rep=False, with_mse=False) -> dict:
        """slate_module.py:198-241 (SLATE branch). Gradients are produced by SLATE.update / backward()."""
        obs = self._need(obs)
        B = obs.shape[0]
        noise, self._injected_noise = self._injected_noise, None
        # fresh tensors, as the reference returns: the engine's metrics buffer is overwritten by the next forward
        m = self.engine.forward(obs, self._tau, self.training, self._next_seed(), noise).clone()
        if self._use_bcdec:        # slate_module.py:218-225
            metrics = {"loss": m[2], "mse": m[0].detach(), "ari": 0}
        else:
            metrics = {
                "loss": m[2],
                "dvae_mse": m[0],
                "cross_entropy": m[1],
                "tau": torch.Tensor([self._tau]),
            }
        if masks is not None:
            from ..utils.tools import calculate_ari
            import numpy as np
            attns = self._attns_image(B)
            fg_mask = 1 - masks[:, -1].unsqueeze(1)
            attns = torch.cat([attns * fg_mask, fg_mask], dim=1)
            if self._use_bcdec:     # the SLATE branch of the reference computes but does not report ari (slate_module.py:231)
                metrics["ari"] = float(np.mean(calculate_ari(masks, attns)))
        if with_mse and not self._use_bcdec:      # slate_module.py:234-237: autoregressive reconstruction error
            metrics = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in metrics.items()}
            self.engine.generate()
            metrics["mse"] = self.engine.metrics[4].clone()
        if with_rep:
            if self._use_bcdec:
                raise NotImplementedError("with_rep is not available with use_bcdec on the HIP backend (the dVAE forward is skipped)")
            E, V = self._enc_size, self._vocab_size
            z = self.engine.tensor("z_st" if self._hard else "z", (B, E, E, V)).permute(0, 3, 1, 2)
            return metrics, z
        return metrics

    def backward(self):
        self.engine.backward()

    def _gen_imgs(self):
        """slate_module.py:163-179 on the slots of the last forward: [B,3,S,S]"""
        B, S = self.engine._keep[0].shape[0], self._obs_size
        self.engine.generate()
        return self.engine.tensor("recon", (B, S, S, 4))[..., :3].permute(0, 3, 1, 2).clone()

    def get_samples(self, obs) -> dict:
        """slate_module.py:243-261: [obs | reconstruction | autoregressive reconstruction | whitened attention maps]"""
        from ..utils.tools import for_viz, visualize
        obs = self._need(obs)
        B, S = obs.shape[0], self._obs_size
        was = self.training
        self.eval()
        self.engine.forward(obs, self._tau, False, self._next_seed(), None)
        self.train(was)
        recon = self.engine.tensor("recon", (B, S, S, 4))[..., :3].permute(0, 3, 1, 2).clone()   # dVAE or broadcast-decoder reconstruction
        attns = self._attns_image(B).clone()
        attns = obs.unsqueeze(1) * attns + (1.0 - attns)
        if self._use_bcdec:
            return {"samples": for_viz(visualize([obs, recon, attns]))}
        return {"samples": for_viz(visualize([obs, recon, self._gen_imgs(), attns]))}

    def load_state_dict(self, state_dict, strict=True):
        sd = {k: v for k, v in state_dict.items()}
        out = super().load_state_dict(sd, strict=strict)
        if self.engine is not None and getattr(self, "_frozen", False):
            self.engine.freeze_weights(True)        # new weights: the derived images are rebuilt once at the next call
        return out


class _EncodeGrad(torch.autograd.Function):
    """Keeps the slots of SLATE_Module.forward attached for a downstream loss: backward hands d loss / d slots to
    ocrl_slate_encode_backward, which fills the flat gradient buffer (encoder tensors) from the activations the encode() call saved."""

    @staticmethod
    def forward(ctx, hook, slots, module, generation):
        ctx.module, ctx.generation = module, generation
        return slots.view_as(slots)

    @staticmethod
    def backward(ctx, g):
        eng = ctx.module.engine
        if eng.encode_generation != ctx.generation:
            raise RuntimeError("ocrl_amd: the encoder ran again before backward(); only the most recent model(obs) call can be differentiated")
        eng.encode_backward(g.contiguous())
        ctx.module._publish_encoder_grads()
        return torch.zeros(1, device=g.device), None, None, None


class FusedAdam:
    """torch.optim.Adam surface (param_groups / state_dict) over the library's fused clip+Adam kernel
    (reference optimiser: ocrs/slate/slate.py:19-34; defaults betas=(0.9,0.999), eps=1e-8, no weight decay)."""

    def __init__(self, module, groups):
        self._module = module
        self.param_groups = []
        for g in groups:
            self.param_groups.append(dict(params=list(g["params"]), lr=g["lr"], betas=(0.9, 0.999), eps=1e-8, weight_decay=0,
                                          amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False,
                                          fused=None, decoupled_weight_decay=False))

    def zero_grad(self, set_to_none=False):
        if self._module.engine is not None:
            self._module.engine.flat_g.zero_()

    def step(self, clip=0.0, grad_scale=1.0):
        lrs = [g["lr"] for g in self.param_groups]
        self._module.engine.clip_adam(lrs, clip, grad_scale)

    def _indexed(self):
        idx, out = 0, []
        for g in self.param_groups:
            ids = list(range(idx, idx + len(g["params"])))
            idx += len(ids)
            out.append(ids)
        return out

    def state_dict(self):
        eng = self._module.engine
        named = {id(p): n for n, p in self._module.named_parameters()}
        spec = {p.name: p for p in eng.params} if eng is not None else {}
        state, groups = {}, []
        for g, ids in zip(self.param_groups, self._indexed()):
            groups.append({**{k: v for k, v in g.items() if k != "params"}, "params": ids})
            for i, p in zip(ids, g["params"]):
                n = named[id(p)]
                if n in spec and eng is not None and eng.adam_step > 0:
                    q = spec[n]
                    state[i] = {"step": torch.tensor(float(eng.adam_step)), "exp_avg": eng.view(eng.flat_m, q).clone(),
                                "exp_avg_sq": eng.view(eng.flat_v, q).clone()}
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        eng = self._module.engine
        if eng is None:         # Base.load() before .to(device) (ocrs/base.py:83-88 callers): applied when the flat buffers exist
            self._module._pending_opt = (self, sd)
            for g, sg in zip(self.param_groups, sd["param_groups"]):
                g["lr"] = sg["lr"]
            return
        named = {id(p): n for n, p in self._module.named_parameters()}
        spec = {p.name: p for p in eng.params}
        for g, sg, ids in zip(self.param_groups, sd["param_groups"], self._indexed()):
            g["lr"] = sg["lr"]
            for i, p in zip(ids, g["params"]):
                st = sd["state"].get(i)
                if st is None:
                    continue
                q = spec[named[id(p)]]
                eng.view(eng.flat_m, q).copy_(st["exp_avg"].to(eng.device))
                eng.view(eng.flat_v, q).copy_(st["exp_avg_sq"].to(eng.device))
                eng.adam_step = int(float(st["step"]))


class SLATE(Base):
    def __init__(self, ocr_config, env_config) -> None:
        self._module = SLATE_Module(ocr_config, env_config)
        super().__init__(ocr_config, env_config)
        lr = self._config.learning
        self._opt = FusedAdam(self._module, [
            {"params": self._module.get_dvae_params(), "lr": lr.lr_dvae},
            {"params": self._module.get_sa_params(), "lr": lr.lr_enc},
            {"params": self._module.get_tfdec_params(), "lr": lr.lr_dec},
        ])
        self._world = None

    def __call__(self, obs, with_attns=False, with_masks=False):
        return self._module(obs, with_attns, with_masks)

    def get_loss(self, obs, masks, with_rep=False, with_mse=False) -> dict:
        """ocrs/slate/slate.py:39-51"""
        out = self._module.get_loss(obs, masks, with_rep, with_mse)
        metrics, rep = out if with_rep else (out, None)
        metrics.update({
            "lr_dvae": torch.Tensor([self._opt.param_groups[0]["lr"]]),
            "lr_enc": torch.Tensor([self._opt.param_groups[1]["lr"]]),
            "lr_dec": torch.Tensor([self._opt.param_groups[2]["lr"]]),
        })
        return (metrics, rep) if with_rep else metrics

    def update(self, obs, masks, step: int) -> dict:
        """ocrs/slate/slate.py:53-69 + ocrs/base.py:60-74: schedules -> loss -> backward ->
        [gradient all-reduce over RCCL when torch.distributed is initialised] -> inf-norm clip -> Adam."""
        self._module.update_tau(step)
        # noise / dropout streams are a function of (seed, rank, global step): a resumed run replays nothing.  Training steps take the
        # even slots of the counter; get_loss() calls outside update() (validation passes, samples) continue on the odd ones
        self._module._step_seed = int(step) * 2
        lr = self._config.learning
        warm = linear_warmup(step, 0, 1, 0, lr.lr_warmup_steps)
        decay = math.exp(step / lr.lr_half_life * math.log(0.5))
        self._opt.param_groups[0]["lr"] = lr.lr_dvae
        self._opt.param_groups[1]["lr"] = decay * warm * lr.lr_enc
        self._opt.param_groups[2]["lr"] = decay * warm * lr.lr_dec
        metrics = self.get_loss(obs, masks)
        self._module.backward()
        scale = allreduce_grads_(self._module.engine.flat_g)     # one flat fp32 buffer; the mean is folded into the clip kernel
        clip = lr.clip if hasattr(lr, "clip") else 0.0
        self._opt.step(clip, scale)
        metrics["norm"] = self._module.engine.metrics[3] * scale
        return metrics

    def to(self, device) -> None:
        self._module.to(device)
