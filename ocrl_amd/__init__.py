"""ocrl_amd — MI355X-native SLATE / Slot-Attention pre-training path behind the reference's
``ocrs`` encoder API.  The compute lives in ``libocrl_hip.so`` (hand-written HIP for gfx950,
C ABI in include/ocrl_hip.h); PyTorch-ROCm only provides device memory, streams and
torch.distributed.  There is no CPU fallback: every compute call raises if the library or the
GPU is missing."""
__version__ = "0.1.0"
