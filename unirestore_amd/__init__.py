"""unirestore_amd — MI355X-native (gfx950) implementation of UniRestore's diffusion-prior restoration path.

Host side: Python on PyTorch-ROCm (memory, streams, torch.distributed).  Compute: hand-written HIP kernels in
libunirestore_hip.so behind the C ABI of include/unirestore_hip.h.  No CPU / eager fallback exists.
"""
__version__ = "0.1.0"
