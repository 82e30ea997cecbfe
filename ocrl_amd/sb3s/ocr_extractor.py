"""Features extractor of the RL side (sb3s/ocr_extractor.py:11-45): observations -> encoder slots -> pooling vector.  With
stable_baselines3 installed it subclasses BaseFeaturesExtractor as the reference does; without it (this image) it is a plain
nn.Module with the same constructor arguments, attributes (``features_dim``) and ``forward``."""
import torch
from torch import nn

from .. import ocrs, poolings

try:                                                          # optional dependency, exactly as in the reference when present
    from stable_baselines3.common.torch_layers import BaseFeaturesExtractor as _BaseExtractor
except Exception:                                             # pragma: no cover - stable_baselines3 is not in this image
    class _BaseExtractor(nn.Module):
        def __init__(self, observation_space, features_dim: int = 0):
            super().__init__()
            self._observation_space = observation_space
            self._features_dim = features_dim

        @property
        def features_dim(self) -> int:
            return self._features_dim


class OCRExtractor(_BaseExtractor):
    def __init__(self, observation_space, config=None):
        ocr = getattr(ocrs, config.ocr.name)(config.ocr, config.env)
        rep_dim = getattr(poolings, config.pooling.name + "_Module")(ocr.rep_dim, ocr.num_slots, config.pooling).rep_dim
        super().__init__(observation_space, rep_dim)
        self._num_envs = config.num_envs
        self._ocr = ocr
        ck = getattr(config.pooling, "ocr_checkpoint", None)
        path = getattr(ck, "local_file", "") if ck is not None else ""
        self._ocr_pretraining = bool(path)
        if path:
            self._ocr.load(torch.load(path, map_location="cpu", weights_only=True))
        self._ocr.to(config.device)                           # get_ocr(..., config.device) in the reference (utils/tools.py)
        self._ocr.eval()
        self._pooling = getattr(poolings, config.pooling.name + "_Module")(ocr.rep_dim, ocr.num_slots, config.pooling)

    def forward(self, observations):
        with torch.no_grad():                                  # the encoder is frozen on this path (poolings/base.py:53)
            slots = self._ocr(observations)
        return self._pooling(slots.detach())
