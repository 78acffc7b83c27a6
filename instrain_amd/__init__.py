"""instrain_amd -- MI355X-native implementation of the `inStrain profile` hot path.

Only what the path needs lives here:
  csrc/       HIP kernels (gfx950) + host BAM front end + the C ABI (include/instrain_amd.h)
  _lib.py     ctypes binding (no CPU fallback)
  engine.py   Context / Batch / BamFile objects over the ABI
  profile/    host-side mirror of the reference's inStrain.profile interface for this path
"""
import os as _os

# A pipe keeps up to a dozen HIP streams busy at once (copy-in, two pass queues, copy-out, one queue per finisher thread).  The HIP
# runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default) and kernels of streams that share a queue run
# one after the other: with 4 queues the finishers' chains of short kernels waited behind the pileup and copy kernels (C5: 129 ->
# 110 ms per pass with 16).  Read by the runtime when it initialises, i.e. before the first HIP call of the process -- so it is set
# at import, and only when the user has not chosen a value.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

__version__ = "0.1.0"
