"""MI355X-native local bundle adjustment + ORB Hamming matching (ORB-SLAM3 hot path).

Product code: ``csrc/`` (HIP kernels + C-ABI, built into ``csrc/liborbslam3_hip.so``),
``capi`` (ctypes mirror of include/orbslam3_hip.h), ``lba`` / ``orb`` (thin Python
drivers over the C-ABI), ``synth`` (synthetic inputs).  The CPU oracle lives in
``/oracle`` and is never imported from this package.
"""
__version__ = "0.1.0"
