"""instrain_amd -- MI355X-native implementation of the `inStrain profile` hot path.

Only what the path needs lives here:
  csrc/       HIP kernels (gfx950) + host BAM front end + the C ABI (include/instrain_amd.h)
  _lib.py     ctypes binding (no CPU fallback)
  engine.py   Context / Batch / BamFile objects over the ABI
  profile/    host-side mirror of the reference's inStrain.profile interface for this path
"""
__version__ = "0.1.0"
