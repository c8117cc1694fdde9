"""MI355X-native (gfx950) conv autoencoder train / reconstruct path behind the reference's trainer surface.

Product code: the HIP library (csrc/, C-ABI in include/uad_hip.h) + this thin Python host side.  There is NO CPU
fallback: importing `_lib` without libuad_hip.so, or calling it without a GPU, raises.
"""
__version__ = '0.1.0'
