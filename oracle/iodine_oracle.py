"""CPU oracle for the IODINE pre-training step (TEST INFRASTRUCTURE ONLY).

Functional PyTorch-CPU fp32 restatement of the reference's IODINE path:
  ocrs/iodine/iodine_module.py:79-252   Iodine_Module._forward (ELBO, in-graph gradients, 17-channel encoding)
  ocrs/iodine/iodine_module.py:254-271  forward / get_loss
  ocrs/iodine/iodine_module.py:307-330  the non-affine "layernorm" (3-D: unbiased std; 5-D: population std; eps on std)
  ocrs/iodine/iodine_module.py:333-373  Decoder (spatial broadcast + coords, 4 x conv3x3 ELU, conv3x3 -> 4)
  ocrs/iodine/iodine_module.py:376-435  RefinementNetwork (4 x conv3x3 stride 2 ELU, avg-pool, Linear ELU ELU, LSTMCell, 2 heads)
  ocrs/base.py:60-74                    update(): clip_grad_norm_(5.0, 2.0) + Adam(lr), one parameter group
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product path never does.
Parity is pinned by tests/golden/make_golden.py, which asserts this restatement equals the imported reference on the same
weights, images and noise and writes tests/golden/iodine_*.npz.

Reference quirks reproduced on purpose:
  * the LSTM outputs are bound as (c, h) = lstm(x, hidden) (:418), so the update heads read the CELL state while
    the tuple handed to the next iteration is still (h_1, c_1) in LSTMCell order;
  * the MLP applies ELU and the caller applies ELU again (:415 + :493);
  * gradients fed to the refinement network are detached (:138-143): no second-order terms reach the loss.
"""
import math
import types
import zlib

import numpy as np
import torch
import torch.nn.functional as F


def default_cfg(**over):
    c = types.SimpleNamespace(
        obs_size=64, obs_channels=3, slot_size=64, num_iterations=5, num_slots=6, sigma=0.35, beta=1.0, layer_norm=True,
        ref_cnn_hidden=64, ref_mlp_hidden=256, ref_cnn_layers=4, ref_cnn_kernel=3, ref_cnn_stride=2,
        dec_cnn_hidden=64, dec_cnn_layers=4, dec_cnn_kernel=3, lr=3e-4, clip=5.0, clip_norm_type=2.0)
    for k, v in over.items():
        if not hasattr(c, k):
            raise KeyError(k)
        setattr(c, k, v)
    return c


ENC_CHANNELS = 17      # image 3, means 3, mask 1, mask_logits 1, mask_posterior 1, grad_means 3, grad_mask 1, likelihood 1, leave-one-out 1, coords 2


def param_shapes(cfg):
    """(name, shape, trainable) in the reference's ``_module.parameters()`` order (iodine_module.py:44-78)."""
    L, Hc, Hm, ks = cfg.slot_size, cfg.ref_cnn_hidden, cfg.ref_mlp_hidden, cfg.ref_cnn_kernel
    # nn.Module yields its own Parameters first, then those of the sub-modules in registration order
    out = [("slot_mean_init", (1, 1, L), True), ("slot_logsig_init", (1, 1, L), True), ("slot_init", (1, 1, L), False)]
    cin = ENC_CHANNELS
    for i in range(cfg.ref_cnn_layers):
        out += [(f"refine.mlc.layers.{i}.weight", (Hc, cin, ks, ks), True), (f"refine.mlc.layers.{i}.bias", (Hc,), True)]
        cin = Hc
    out += [("refine.mlp.layers.0.weight", (Hm, Hc), True), ("refine.mlp.layers.0.bias", (Hm,), True)]
    out += [("refine.lstm.weight_ih", (4 * Hm, Hm + 4 * L), True), ("refine.lstm.weight_hh", (4 * Hm, Hm), True),
            ("refine.lstm.bias_ih", (4 * Hm,), True), ("refine.lstm.bias_hh", (4 * Hm,), True)]
    out += [("refine.mean_update.weight", (L, Hm), True), ("refine.mean_update.bias", (L,), True),
            ("refine.logsig_update.weight", (L, Hm), True), ("refine.logsig_update.bias", (L,), True)]
    cin, Hd, kd = L + 2, cfg.dec_cnn_hidden, cfg.dec_cnn_kernel
    for i in range(cfg.dec_cnn_layers):
        out += [(f"decoder.mlc.layers.{i}.weight", (Hd, cin, kd, kd), True), (f"decoder.mlc.layers.{i}.bias", (Hd,), True)]
        cin = Hd
    out += [("decoder.conv.weight", (4, Hd, kd, kd), True), ("decoder.conv.bias", (4,), True)]
    return out


def formula_tensor(name, shape, scale):
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    return torch.from_numpy((rs.standard_normal(shape) * scale).astype(np.float32))


def formula_params(cfg):
    """Deterministic, well-conditioned weights (closed form from the name) for parity runs."""
    P = {}
    for name, shape, _ in param_shapes(cfg):
        if name.endswith("bias"):
            P[name] = formula_tensor(name, shape, 0.05)
        elif name in ("slot_mean_init", "slot_logsig_init", "slot_init"):
            P[name] = formula_tensor(name, shape, 0.3)
        else:
            fan_in = int(np.prod(shape[1:]))
            gain = 0.1 if "_update" in name else (0.3 if name == "decoder.conv.weight" else 1.0)    # keeps the 5-iteration refinement tame
            P[name] = formula_tensor(name, shape, gain / math.sqrt(fan_in))
    return P


def coords(S, dtype=torch.float32):
    """[2,S,S]: channel 0 = xx (varies along W), channel 1 = yy (iodine_module.py:452-456, :221-225)."""
    lin = torch.linspace(-1, 1, S, dtype=dtype)
    yy, xx = torch.meshgrid(lin, lin, indexing="ij")
    return torch.stack((xx, yy), dim=0)


def layernorm3(x):
    """(B,K,L): unbiased std over L, eps added to the std (iodine_module.py:313-315,329)."""
    m = x.mean(dim=2, keepdim=True)
    s = x.std(dim=2, keepdim=True)
    return (x - m) / (s + 1e-5)


def layernorm5(x):
    """(B,K,C,H,W): population std over (C,H,W) (iodine_module.py:316-326,329)."""
    m = x.mean(dim=(2, 3, 4), keepdim=True)
    s = torch.sqrt(((x - m) ** 2).mean(dim=(2, 3, 4), keepdim=True))
    return (x - m) / (s + 1e-5)


def decoder(P, slots, cfg):
    """slots [B,K,L] -> recons [B,K,3,S,S], mask_logits [B,K,1,S,S] (iodine_module.py:333-373,438-470)."""
    B, K, L = slots.shape
    S = cfg.obs_size
    x = slots.reshape(B * K, L)[:, :, None, None].expand(B * K, L, S, S)
    x = torch.cat((x, coords(S)[None].expand(B * K, 2, S, S)), dim=1)
    pad = cfg.dec_cnn_kernel // 2
    for i in range(cfg.dec_cnn_layers):
        x = F.elu(F.conv2d(x, P[f"decoder.mlc.layers.{i}.weight"], P[f"decoder.mlc.layers.{i}.bias"], padding=pad))
    x = F.conv2d(x, P["decoder.conv.weight"], P["decoder.conv.bias"], padding=pad)
    return x[:, :3].reshape(B, K, 3, S, S), x[:, 3:].reshape(B, K, 1, S, S)


def refine(P, enc, latent, hidden, cfg):
    """enc [B,K,17,S,S], latent [B,K,4L], hidden None | (h,c) -> mean_delta, logsig_delta, new hidden (iodine_module.py:376-435)."""
    B, K = enc.shape[:2]
    x = enc.reshape(B * K, *enc.shape[2:])
    pad = cfg.ref_cnn_kernel // 2
    for i in range(cfg.ref_cnn_layers):
        x = F.elu(F.conv2d(x, P[f"refine.mlc.layers.{i}.weight"], P[f"refine.mlc.layers.{i}.bias"], stride=cfg.ref_cnn_stride, padding=pad))
    x = x.mean(dim=(2, 3))
    x = F.elu(F.elu(F.linear(x, P["refine.mlp.layers.0.weight"], P["refine.mlp.layers.0.bias"])))
    x = torch.cat((x, latent.reshape(B * K, -1)), dim=1)
    Hm = cfg.ref_mlp_hidden
    if hidden is None:
        h0 = x.new_zeros(B * K, Hm)
        c0 = x.new_zeros(B * K, Hm)
    else:
        h0, c0 = hidden
    gates = F.linear(x, P["refine.lstm.weight_ih"], P["refine.lstm.bias_ih"]) + F.linear(h0, P["refine.lstm.weight_hh"], P["refine.lstm.bias_hh"])
    gi, gf, gg, go = gates.chunk(4, dim=1)
    c1 = torch.sigmoid(gf) * c0 + torch.sigmoid(gi) * torch.tanh(gg)
    h1 = torch.sigmoid(go) * torch.tanh(c1)
    head_in = c1                         # the reference names the LSTMCell outputs (c, h): the heads read the cell state
    md = F.linear(head_in, P["refine.mean_update.weight"], P["refine.mean_update.bias"]).reshape(B, K, -1)
    ld = F.linear(head_in, P["refine.logsig_update.weight"], P["refine.logsig_update.bias"]).reshape(B, K, -1)
    return md, ld, (h1, c1)


def elbo_terms(image, means, logsigs, slots, recons, mask_logits, cfg):
    """ELBO pieces of one iteration (iodine_module.py:92-119)."""
    B = image.shape[0]
    masks = F.softmax(mask_logits, dim=1)
    recons_masked = masks * recons
    recon = recons_masked.sum(dim=1)
    mse = ((image - recon) ** 2).sum() / B
    sig = logsigs.exp()
    kl = (-logsigs + 0.5 * (sig * sig + means * means) - 0.5).sum() / B
    clp = -((image[:, None] - recons) ** 2) / (2 * cfg.sigma ** 2) - math.log(cfg.sigma) - 0.5 * math.log(2 * math.pi)
    pll = torch.logsumexp((masks + 1e-12).log() + clp, dim=1, keepdim=True)
    ll = pll.sum() / B
    elbo = ll - cfg.beta * kl
    return dict(masks=masks, recons_masked=recons_masked, recon=recon, mse=mse, kl=kl, clp=clp, pll=pll, elbo=elbo)


def encoding(image, recons, mask_logits, t, recons_grad, masks_grad, cfg):
    """The 17-channel refinement input (iodine_module.py:145-229)."""
    B, K = recons.shape[:2]
    S = cfg.obs_size
    masks, clp, pll = t["masks"], t["clp"], t["pll"]
    ln5 = layernorm5 if cfg.layer_norm else (lambda v: v)
    a_clp = clp.sum(dim=2, keepdim=True)
    like = pll.sum(dim=2, keepdim=True).exp().expand(B, K, 1, S, S)
    a_p = a_clp.exp()
    loo = ((masks * a_p).sum(dim=1, keepdim=True) - masks * a_p) / (1 - masks + 1e-5)
    return torch.cat([
        image[:, None].expand(B, K, 3, S, S), recons, masks, mask_logits, torch.log_softmax(a_clp, dim=1),
        ln5(recons_grad), ln5(masks_grad), ln5(like).detach(), ln5(loo).detach(),
        coords(S)[None, None].expand(B, K, 2, S, S)], dim=2)


def iodine_forward(P, image, eps, cfg, return_all=False):
    """image [B,3,S,S]; eps [I,B,K,L] N(0,1) draws of ``rsample`` in call order -> dict (iodine_module.py:79-252)."""
    B = image.shape[0]
    K, I = cfg.num_slots, cfg.num_iterations
    means = P["slot_mean_init"].repeat(B, K, 1)
    logsigs = P["slot_logsig_init"].repeat(B, K, 1)
    hidden = None
    elbos, trace = [], []
    ln3 = layernorm3 if cfg.layer_norm else (lambda v: v)
    for i in range(I):
        slots = means + logsigs.exp() * eps[i]
        recons, mask_logits = decoder(P, slots, cfg)
        t = elbo_terms(image, means, logsigs, slots, recons, mask_logits, cfg)
        elbos.append(t["elbo"])
        if return_all:
            trace.append(dict(slots=slots, recons=recons, mask_logits=mask_logits, masks=t["masks"], elbo=t["elbo"], means=means, logsigs=logsigs))
        if i < I - 1:
            need = [means, logsigs, recons, t["masks"]]
            if not any(v.requires_grad for v in need):
                raise RuntimeError("iodine oracle: parameters must require grad (the refinement inputs are gradients)")
            g_m, g_s, g_r, g_k = torch.autograd.grad(B * t["elbo"], need, retain_graph=True)
            latent = torch.cat((means, logsigs, ln3(g_m.detach()), ln3(g_s.detach())), dim=-1)
            enc = encoding(image, recons, mask_logits, t, g_r.detach(), g_k.detach(), cfg)
            if return_all:
                trace[-1].update(enc=enc, latent=latent)
            md, ld, hidden = refine(P, enc, latent, hidden, cfg)
            means = means + md
            logsigs = logsigs + ld
    elbo = sum((i + 1) / I * e for i, e in enumerate(elbos))
    out = dict(slots=slots, recon=t["recon"].clamp(0, 1), recons_masked=t["recons_masked"].clamp(0, 1), masks=t["masks"],
               loss=-elbo, mse=t["mse"], kl=t["kl"], means=recons.clamp(0, 1))
    if return_all:
        out["trace"] = trace
    return out


def grad_clip_l2(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_(params, max_norm, 2.0): returns (norm, coefficient) (base.py:65-70)."""
    tot = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(max_norm / (tot + 1e-6), max=1.0)
    return tot, coef


def adam_update(p, g, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8):
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
    p.addcdiv_(m, (v.sqrt() / math.sqrt(bc2)).add_(eps), value=-lr / bc1)


class OracleTrainer:
    """``Iodine.update`` (base.py:60-74): loss -> backward -> L2 clip -> Adam, on plain tensors."""

    def __init__(self, cfg, params):
        self.cfg = cfg
        self.P = {k: v.clone() for k, v in params.items()}
        self.trainable = [n for n, _, tr in param_shapes(cfg) if tr]
        self.m = {n: torch.zeros_like(self.P[n]) for n in self.trainable}
        self.v = {n: torch.zeros_like(self.P[n]) for n in self.trainable}
        self.t = 0

    def loss_and_grads(self, image, eps):
        P = {k: (v.clone().requires_grad_(True) if k in self.trainable else v) for k, v in self.P.items()}
        out = iodine_forward(P, image, eps, self.cfg)
        names = [n for n in self.trainable]
        gs = torch.autograd.grad(out["loss"], [P[n] for n in names], allow_unused=True)
        grads = {n: (g if g is not None else torch.zeros_like(P[n])) for n, g in zip(names, gs)}
        return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}, grads

    def update(self, image, eps):
        out, grads = self.loss_and_grads(image, eps)
        norm, coef = grad_clip_l2(grads, self.cfg.clip)
        self.t += 1
        for n in self.trainable:
            adam_update(self.P[n], grads[n] * coef, self.m[n], self.v[n], self.t, self.cfg.lr)
        out["norm"] = norm
        return out


def make_noise(cfg, B, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(cfg.num_iterations, B, cfg.num_slots, cfg.slot_size, generator=g)
