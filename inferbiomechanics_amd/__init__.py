"""MI355X-native hot path of InferBiomechanics (model fwd/bwd, loss, optimizer step, DDP, DDIM).

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all arithmetic of
the hot path runs in hand-written gfx950 HIP kernels behind the C-ABI of ``include/ib_hip.h``.
"""
__version__ = "0.1.0"
