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
        # get_ocr (utils/tools.py:323-347): without a checkpoint, or with ocr_checkpoint.finetuning, the extractor owns the encoder *module*
        # -- an nn.Module, so its parameters are the policy's and the RL loss trains them through the slots; with a checkpoint and no
        # finetuning it owns the wrapper object, whose parameters no optimiser sees: a frozen, pre-trained encoder
        ck = getattr(config.pooling, "ocr_checkpoint", None)
        path = getattr(ck, "local_file", "") if ck is not None else ""
        if not path and ck is not None and getattr(ck, "run_id", "") != "":
            raise RuntimeError("ocr_checkpoint.run_id needs wandb; download the file and set ocr_checkpoint.local_file")
        self._ocr_pretraining = bool(path)
        if path:
            ocr.load(torch.load(path, map_location="cpu", weights_only=True))
        ocr.to(config.device)                                 # get_ocr(..., config.device)
        self._trainable = (not path) or bool(getattr(ck, "finetuning", False))
        mod = getattr(ocr, "_module", None)
        if self._trainable and (mod is None or not hasattr(mod, "finetune_through_slots") or getattr(mod, "_use_cnn_feat", False)):
            raise NotImplementedError("training the encoder through the RL loss is built for the SLATE / Slot-Attention slots; give a "
                                      "pre-trained checkpoint (pooling.ocr_checkpoint.local_file) without finetuning for this encoder")
        if self._trainable:
            mod.finetune_through_slots = True                 # the slots stay attached: d loss / d slots -> ocrl_slate_encode_backward
            self._ocr = mod                                    # registered as a sub-module: its parameters are the extractor's
        else:
            ocr.eval()
            if hasattr(mod, "freeze_weights"):
                mod.freeze_weights(True)                       # constant weights: no re-packing per call
            self._ocr = ocr
        self._pooling = getattr(poolings, config.pooling.name + "_Module")(ocr.rep_dim, ocr.num_slots, config.pooling)

    def forward(self, observations):
        if self._trainable:
            return self._pooling(self._ocr(observations))     # sb3s/ocr_extractor.py:45
        with torch.no_grad():                                  # nothing trains the wrapper's parameters: skip the backward bookkeeping
            slots = self._ocr(observations)
        return self._pooling(slots.detach())
