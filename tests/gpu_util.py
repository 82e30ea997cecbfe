"""Helpers shared by the GPU parity tests."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")


def log(msg):
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "parity_log.txt"), "a") as f:
        f.write(msg + "\n")
    print(msg)
    sys.stdout.flush()


def relerr(a, b, floor=0.0):
    """max |a-b| / max(max|b|, floor)"""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    den = max(b.abs().max().item(), floor, 1e-30)
    return ((a - b).abs().max().item()) / den


# Parameters whose exact gradient is identically zero: a shift of norm_slots.bias moves every slot's query by the same vector, i.e. every
# slot's logit at a position by the same amount, and the softmax over slots cancels it.  Both sides then hold pure rounding residue of a
# sum whose terms cancel, so they are graded on a coarser floor (1e-3 x the largest gradient instead of 1e-5 x).
EXACT_ZERO_GRAD = ("_slotattn.slot_attention.norm_slots.bias",)


def grad_floor(name, gmax):
    return (1e-3 if name in EXACT_ZERO_GRAD else 1e-5) * gmax


def dims_from_cfg(cfg):
    from types import SimpleNamespace
    return SimpleNamespace(obs_size=cfg.obs_size, obs_channels=cfg.obs_channels, vocab_size=cfg.vocab_size, d_model=cfg.d_model,
                           cnn_hidden=cfg.cnn_hidden, num_slots=cfg.num_slots, num_iterations=cfg.num_iterations,
                           slot_size=cfg.slot_size, mlp_hidden=cfg.mlp_hidden, num_dec_blocks=cfg.num_dec_blocks,
                           num_dec_heads=cfg.num_dec_heads, dropout=cfg.dropout, use_bcdec=cfg.use_bcdec, hard=cfg.hard,
                           num_slot_heads=int(getattr(cfg, "num_slot_heads", 1)))


def load_params(engine, P):
    """oracle param dict (reference state_dict names) -> engine flat buffer"""
    for p in engine.params:
        engine.view(engine.flat_p, p).copy_(P[p.name].to(engine.device).reshape(p.shape))
    torch.cuda.synchronize()


def reference_style_config(cfg, use_cnn_feat=False):
    """(ocr_config, env_config) namespaces in the reference's layout (configs/ocr/slate.yaml) for an oracle cfg"""
    from types import SimpleNamespace as NS
    ocr = NS(name="SLATE", tau_start=cfg.tau_start, tau_final=cfg.tau_final, tau_steps=cfg.tau_steps, hard=cfg.hard, use_cnn_feat=use_cnn_feat,
             use_bcdec=cfg.use_bcdec, dvae=NS(vocab_size=cfg.vocab_size, d_model=cfg.d_model), cnn=NS(hidden_size=cfg.cnn_hidden),
             slotattr=NS(num_iterations=cfg.num_iterations, num_slots=cfg.num_slots, num_slot_heads=cfg.num_slot_heads, slot_size=cfg.slot_size,
                         mlp_hidden_size=cfg.mlp_hidden, pos_channels=4),
             tfdec=NS(num_dec_blocks=cfg.num_dec_blocks, num_dec_heads=cfg.num_dec_heads),
             learning=NS(lr_half_life=cfg.lr_half_life, lr_dvae=cfg.lr_dvae, lr_enc=cfg.lr_enc, lr_dec=cfg.lr_dec, lr_warmup_steps=cfg.lr_warmup_steps,
                         dropout=cfg.dropout, clip=cfg.clip))
    return ocr, NS(obs_size=cfg.obs_size, obs_channels=cfg.obs_channels)


def build_wrapper(cfg, P, use_cnn_feat=False, device="cuda:0"):
    """ocrs.SLATE (the drop-in wrapper) holding the oracle's closed-form weights, on the GPU, eval mode"""
    from ocrl_amd import ocrs
    ocr, env = reference_style_config(cfg, use_cnn_feat)
    model = ocrs.SLATE(ocr, env)
    sd = model._module.state_dict()
    model._module.load_state_dict({k: (P[k] if k in P else sd[k]) for k in sd})
    model.to(device)
    model.eval()
    return model


# ---------------------------------------------------------------------------------------------------------------------------
# ReLU-mask-matched gradients.  Two fp32 evaluations of the step can disagree on the sign of a pre-activation that lies inside
# rounding noise of zero; the unit's whole contribution to the upstream weight gradients then appears or vanishes (measured:
# one flip at |pre| = 4.9e-8 in a 64x64 case = 2.7e-4 of a conv weight gradient's max).  That is a property of ReLU in fp32, not
# of either implementation, so the gradient comparison holds the ReLU decisions fixed: the oracle is evaluated in fp64 with the
# masks of the HIP forward (read from its saved activations), and every tensor must then agree tightly.
# ---------------------------------------------------------------------------------------------------------------------------
def hip_relu_acts(eng, cfg, B, images=None):
    """post-ReLU activations of the HIP forward (zero where the unit is closed), in the order and tensor layout of the oracle's F.relu
    calls (slate_loss); `images` restricts the batch dimension (a slice) before the copy to the host"""
    S, E = cfg.obs_size, cfg.obs_size // 4
    T, N, K, I, H, d = E * E, S * S, cfg.num_slots, cfg.num_iterations, cfg.mlp_hidden, cfg.d_model
    sl = images if images is not None else slice(None)
    nchw = lambda t: t[sl].permute(0, 3, 1, 2).cpu()
    out = []
    if cfg.use_bcdec:       # the oracle (like the reference) still runs the dVAE here; the HIP path skips that dead work: plain ReLU for those calls
        out += [None] * 16
    else:
        for i in range(7):
            out.append(nchw(eng.tensor(f"dvae_enc{i}", (B, E, E, 64))))
        for name, hh, c in (("dvae_dec0", E, 64), ("dvae_dec1", E, 64), ("dvae_dec2", E, 64), ("dvae_dec3", E, 64), ("dvae_dec4", E, 256),
                            ("dvae_dec6", 2 * E, 64), ("dvae_dec7", 2 * E, 64), ("dvae_dec8", 2 * E, 64), ("dvae_dec9", 2 * E, 256)):
            out.append(nchw(eng.tensor(name, (B, hh, hh, c))))
    for name in ("enc1", "enc2", "enc3"):
        out.append(nchw(eng.tensor(name, (B, S, S, 64))))
    out.append(eng.tensor("sa_mlp_hidden", (B, N, 64))[sl].cpu())
    nh = int(getattr(cfg, "num_slot_heads", 1))
    ld = 10 * cfg.slot_size + H + 3 * 64 * nh + ((nh + 3) & ~3)               # kernels.h sa_save_layout
    sv = eng.tensor("sa_save", (B, I, K, ld))[..., 10 * cfg.slot_size:10 * cfg.slot_size + H]
    for t in range(I):
        out.append(sv[sl, t].cpu())
    if cfg.use_bcdec:
        ks = slice(None) if images is None else slice((images.start or 0) * K, (images.stop or B) * K)
        for name in ("bc_c1", "bc_c2", "bc_c3"):
            out.append(eng.tensor(name, (B * K, S, S, 64))[ks].permute(0, 3, 1, 2).cpu())
    else:
        for b in range(cfg.num_dec_blocks):
            out.append(eng.tensor(f"blk{b}.ffn_hidden", (B, T, 4 * d))[sl].cpu())
    return out


def hip_relu_masks(eng, cfg, B):
    """bool masks of every ReLU of the HIP forward, in the order and tensor layout of the oracle's F.relu calls (slate_loss)"""
    return [None if a is None else a > 0 for a in hip_relu_acts(eng, cfg, B)]


class ForcedRelu:
    """F.relu replaced by x * mask[i] for the i-th call (None = plain ReLU); counts the units whose natural mask differs"""

    def __init__(self, masks):
        import torch.nn.functional as F
        self.F, self.masks, self.i, self.flips, self.units, self.min_flipped = F, masks, 0, 0, 0, []
        self._orig = F.relu

    def __enter__(self):
        def relu(x, inplace=False):
            m = self.masks[self.i] if self.i < len(self.masks) else None
            self.i += 1
            if m is None:
                return self._orig(x)
            assert m.shape == x.shape, (self.i - 1, m.shape, x.shape)
            nat = x > 0
            diff = nat != m
            n = int(diff.sum())
            self.units += x.numel()
            if n:
                self.flips += n
                self.min_flipped.append(float(x.detach()[diff].abs().max()))
            return x * m.to(x.dtype)
        self.F.relu = relu
        return self

    def __exit__(self, *a):
        self.F.relu = self._orig
        assert self.i == len(self.masks), (self.i, len(self.masks))


def assert_knife_edge(fr, tag=""):
    """Mask matching may excuse rounding ties and nothing else: the ReLU decisions imposed on the fp64 run may differ from fp64's own
    only in a handful of units whose pre-activation lies inside fp32 rounding noise of zero.  A kernel that takes a wrong decision at a
    non-negligible pre-activation, or many of them, fails here instead of being carried into the oracle."""
    limit = max(4, int(1e-6 * fr.units))
    assert fr.flips <= limit, f"{tag}: {fr.flips} of {fr.units} ReLU decisions of the HIP forward differ from the fp64 oracle's own (limit {limit})"
    if fr.flips:
        worst = max(fr.min_flipped)
        assert worst <= 1e-5, f"{tag}: a flipped ReLU decision at |pre-activation| = {worst:.2e} (> 1e-5) is not a rounding tie"


def mask_matched_fp64_grads(cfg, P, obs, noise, step, masks, drop_masks=None):
    """fp64 run of the oracle with the given ReLU decisions; returns (trainer with .P[name].grad, ForcedRelu stats)"""
    from oracle import slate_oracle as O
    P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
    tr = O.OracleTrainer(cfg, P64)
    dm = None if drop_masks is None else {k: v.double() for k, v in drop_masks.items()}
    with ForcedRelu(masks) as fr:
        tr.loss_and_grads(obs.double(), {k: v.double() for k, v in noise.items()}, step, dm)
    return tr, fr
