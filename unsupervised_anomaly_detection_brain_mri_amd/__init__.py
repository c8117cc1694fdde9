"""MI355X-native (gfx950) conv autoencoder train / reconstruct path behind the reference's trainer surface.

Product code: the HIP library (csrc/, C-ABI in include/uad_hip.h) + this thin Python host side.  There is NO CPU
fallback: importing `_lib` without libuad_hip.so, or calling it without a GPU, raises.
"""
import os as _os

# One process per GPU under a launcher (WORLD_SIZE > 1): the engine keeps two streams busy (compute + its side stream for slab reductions / BN
# finalizes / the weight repack) and the data-parallel layer adds the process group's.  HIP maps streams onto at most GPU_MAX_HW_QUEUES hardware
# queues (default 4); an RCCL communicator created before the handle takes queue slots first, the handle's two streams then share ONE hardware
# queue and serialise: +17 % per step with no collective issued at all (profiles/r04_k_rccl_hw_queues.log).  Eight queues keep them apart.  The
# HIP runtime reads the variable when it initialises, so this has to happen before the first HIP call of the process: import this package (or set
# the variable) before touching torch.cuda.  Single-process runs keep the runtime's default (the configuration every committed profile was taken in).
if int(_os.environ.get('WORLD_SIZE', '1') or 1) > 1 or _os.environ.get('UAD_BENCH_REHEARSAL'):
    _os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

__version__ = '0.1.0'
