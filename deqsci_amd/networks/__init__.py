"""Denoiser plugins (plain torch.nn on PyTorch-ROCm / MIOpen) with the reference's class names,
constructor arguments and state-dict keys, so reference checkpoints load unchanged."""
from .ffdnet import FFDNet, IntermediateDnCNN  # noqa: F401
from .simplecnn import DnCNN, RealSNConv2d  # noqa: F401
