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
