"""MI355X-native (gfx950) conv autoencoder train / reconstruct path behind the reference's trainer surface.

Product code: the HIP library (csrc/, C-ABI in include/uad_hip.h) + this thin Python host side.  There is NO CPU
fallback: importing `_lib` without libuad_hip.so, or calling it without a GPU, raises.
"""
import os as _os

# The engine keeps two streams busy (compute + its side stream for slab reductions / BN finalizes / the weight repack) and the data-parallel
# layer adds the process group's.  HIP maps streams onto at most GPU_MAX_HW_QUEUES hardware queues (default 4); an RCCL communicator created
# BEFORE the handle takes queue slots first, the handle's two streams then share ONE hardware queue and serialise: measured +17 % per step with
# no collective issued at all (profiles/r04_k_rccl_hw_queues.log).  Eight queues keep them apart.  Read by the HIP runtime when it initialises,
# so this has to happen before the first HIP call of the process: import this package (or set the variable) before touching torch.cuda.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

__version__ = '0.1.0'
