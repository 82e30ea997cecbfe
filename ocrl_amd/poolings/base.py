"""Pooling wrapper API of the reference (poolings/base.py:5-89), kept at the Python surface: owns the OCR wrapper and the pooling
``_module``; the arithmetic of both is in libocrl_hip."""
import torch


class Base:
    def __init__(self, ocr, config) -> None:
        self._ocr = ocr
        self._config = config
        self._learn_aux_loss = config.learn_aux_loss
        self._learn_downstream_loss = config.learn_downstream_loss
        if self._learn_downstream_loss:
            # poolings/base.py:53-55: the slots reach the pooling head undetached.  The encoder's forward is a library call, not a torch
            # graph: its module keeps the slots attached through an autograd function that routes d loss / d slots into
            # ocrl_slate_encode_backward (ocrs/slate.py::_EncodeGrad); set_zero_grad() / do_step() then act on the flat buffers
            mod = getattr(ocr, "_module", None)
            if mod is None or not hasattr(mod, "finetune_through_slots") or getattr(mod, "_use_cnn_feat", False):
                raise NotImplementedError("learn_downstream_loss=True is built for the SLATE / Slot-Attention encoder's slots (not use_cnn_feat, not IODINE)")
            mod.finetune_through_slots = True
        self._load_ocr()
        self.rep_dim = self._module.rep_dim
        if hasattr(self._config, "learning") and hasattr(self._config.learning, "lr"):
            self._opt = torch.optim.Adam(self._module.parameters(), lr=config.learning.lr)

    def _load_ocr(self):
        """poolings/base.py:24-29: restore the pre-trained encoder when a checkpoint is configured (local files only: no wandb here)"""
        ck = getattr(self._config, "ocr_checkpoint", None)
        path = getattr(ck, "local_file", "") if ck is not None else ""
        if path:
            self._ocr.load(torch.load(path, map_location="cpu", weights_only=True))
        elif ck is not None and getattr(ck, "run_id", "") != "":
            raise RuntimeError("ocr_checkpoint.run_id needs wandb; download the file and set ocr_checkpoint.local_file")

    def set_zero_grad(self):
        if hasattr(self, "_opt"):
            self._opt.zero_grad()
        self._ocr.set_zero_grad()

    def do_step(self):
        if hasattr(self, "_opt"):
            self._opt.step()
        self._ocr.do_step()

    def __call__(self, obs, with_loss=False):
        if self._learn_aux_loss and with_loss:
            metrics, state = self._ocr.get_loss(obs, with_rep=True)
            metrics["aux_loss"] = metrics.pop("loss")
        else:
            state = self._ocr(obs)
            metrics = {}
        # detach if not fine-tuning (poolings/base.py:53-55)
        state = state.detach() if not self._learn_downstream_loss else state
        state = self._module(state)
        return (state, metrics) if with_loss else state

    def train(self) -> None:
        self._module.train()
        self._ocr.train()

    def eval(self) -> None:
        self._module.eval()
        self._ocr.eval()

    def to(self, device) -> None:
        self._module.to(device)
        self._ocr.to(device)

    def get_samples(self, obs) -> dict:
        return self._ocr.get_samples(obs)

    def save(self) -> dict:
        checkpoint = {"pooling_module_state_dict": self._module.state_dict()}
        if hasattr(self, "_opt"):
            checkpoint["pooling_opt_state_dict"] = self._opt.state_dict()
        checkpoint.update(self._ocr.save())
        return checkpoint

    def load(self, checkpoint) -> None:
        self._module.load_state_dict(checkpoint["pooling_module_state_dict"])
        if hasattr(self, "_opt"):
            self._opt.load_state_dict(checkpoint["pooling_opt_state_dict"])
        self._ocr.load(checkpoint)
