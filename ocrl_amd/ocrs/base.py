"""Encoder wrapper API of the reference (ocrs/base.py:8-88), kept verbatim at the Python surface.
Subclasses own ``_module`` (parameters, state_dict) and ``_opt``; the arithmetic is in libocrl_hip."""
import torch


class Base:
    def __init__(self, ocr_config, env_config) -> None:
        self.name = ocr_config.name
        self._config = ocr_config
        self._obs_size = env_config.obs_size
        self._obs_channels = env_config.obs_channels
        # for pooling layer (ocrs/base.py:16-18)
        self.rep_dim = self._module.rep_dim
        self.num_slots = self._module.num_slots

    def __call__(self, obs):
        return self._module(obs)

    def wandb_watch(self, config):
        """wandb.watch(module) in the reference (base.py:30-31); optional here (wandb is not required)."""
        try:
            import wandb
            wandb.watch(self._module, log=getattr(config, "log", None))
        except Exception:
            pass

    def get_loss(self, obs, with_rep=False):
        return self._module.get_loss(obs, with_rep)

    def train(self) -> None:
        self._module.train()
        return None

    def eval(self) -> None:
        self._module.eval()
        return None

    def to(self, device) -> None:
        self._module.to(device)

    def set_zero_grad(self):
        if hasattr(self, "_opt"):
            self._opt.zero_grad()

    def do_step(self):
        if hasattr(self, "_opt"):
            self._opt.step()

    def get_samples(self, obs) -> dict:
        return self._module.get_samples(obs)

    def save(self) -> dict:
        checkpoint = {"ocr_module_state_dict": self._module.state_dict()}
        if hasattr(self, "_opt"):
            checkpoint["ocr_opt_state_dict"] = self._opt.state_dict()
        return checkpoint

    def load(self, checkpoint) -> None:
        self._module.load_state_dict(checkpoint["ocr_module_state_dict"])
        if hasattr(self, "_opt") and "ocr_opt_state_dict" in checkpoint:
            self._opt.load_state_dict(checkpoint["ocr_opt_state_dict"])
