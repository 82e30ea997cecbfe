"""Training services the SLATE path touches (reference: utils/tools.py). wandb / h5py / omegaconf are optional."""
import numpy as np
import torch


def to_device(batch, device):
    """utils/tools.py:182-191"""
    if isinstance(batch, list):
        return [b.to(device) for b in batch]
    if isinstance(batch, dict):
        return {k: v.to(device) for k, v in batch.items()}
    return batch.to(device)


def obs_from_uint8(u8):
    """uint8 [B,H,W,C] on the GPU -> fp32 [B,C,H,W] in [0,1] (utils/datasets.py:17 + the H2D copy of train_ocr.py:52-53), converted by
    the library's kernel: the host ships a quarter of the bytes and never touches the pixels"""
    import ctypes
    from .. import _lib
    if not (u8.is_cuda and u8.dtype == torch.uint8 and u8.dim() == 4):
        raise RuntimeError("obs_from_uint8: expected a uint8 [B,H,W,C] tensor on the GPU")
    u8 = u8.contiguous()
    B, H, W, C = u8.shape
    out = torch.empty(B, C, H, W, dtype=torch.float32, device=u8.device)
    _lib.check(_lib.lib().ocrl_obs_u8_to_f32(_lib.ptr(u8), _lib.ptr(out), B, H, W, C, ctypes.c_void_p(torch.cuda.current_stream(u8.device).cuda_stream)))
    return out


def get_item(x):
    """utils/tools.py:195-199"""
    if not torch.is_tensor(x):
        return x
    if len(x.shape) == 0:
        return x.item()
    return x.detach().cpu().numpy()


def for_viz(x):
    """utils/tools.py:203-205"""
    return np.array(x.clamp(0, 1).permute(0, 2, 3, 1).detach().cpu().numpy() * 255.0, dtype=np.uint8)


def visualize(images):
    """utils/tools.py:209-219: width-concatenate [B,3,H,W] images and [B,K,3,H,W] stacks"""
    viz = []
    for img in images:
        if img.dim() == 4:
            viz.append(img)
        else:
            viz += list(torch.unbind(img, dim=1))
    return torch.cat(viz, dim=-1)


def _adjusted_rand_score(a, b):
    """sklearn.metrics.adjusted_rand_score (pair-counting form) for two integer label vectors"""
    a = np.asarray(a).ravel()
    b = np.asarray(b).ravel()
    n = a.size
    _, ai = np.unique(a, return_inverse=True)
    _, bi = np.unique(b, return_inverse=True)
    cont = np.zeros((ai.max() + 1, bi.max() + 1), dtype=np.int64)
    np.add.at(cont, (ai, bi), 1)
    comb = lambda x: x * (x - 1) // 2
    sum_ij = comb(cont).sum()
    sa, sb = comb(cont.sum(1)).sum(), comb(cont.sum(0)).sum()
    tot = comb(np.int64(n))
    if tot == 0:
        return 1.0
    exp = sa * sb / tot
    mx = 0.5 * (sa + sb)
    if mx == exp:
        return 1.0
    return float((sum_ij - exp) / (mx - exp))


def calculate_ari(true_masks, pred_masks):
    """utils/tools.py:309-320"""
    t = torch.argmax(true_masks.flatten(2), dim=1).cpu().numpy()
    p = torch.argmax(pred_masks.flatten(2), dim=1).cpu().numpy()
    return [_adjusted_rand_score(t[b], p[b]) for b in range(t.shape[0])]
