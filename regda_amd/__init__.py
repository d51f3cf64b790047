"""regda_amd -- MI355X-native (gfx950) implementation of the RegDA self-training step.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed);
all compute on the path goes through the C ABI of `csrc/librgda_hip.so`
(`include/rgda_hip.h`).  There is NO CPU fallback: importing `regda_amd._lib`
without the built library raises.
"""
__version__ = '0.1.0'
