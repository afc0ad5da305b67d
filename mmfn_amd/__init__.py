"""mmfn_amd: MI355X-native (gfx950) implementation of the MMFN training hot path.

The compute path is hand-written HIP behind a C ABI (include/mmfn_hip.h, mmfn_amd/csrc);
PyTorch supplies device memory, streams and torch.distributed only.
"""
__version__ = "0.1.0"
