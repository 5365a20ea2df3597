"""MI355X-native hot path of hvak/visual-underwater-slam.

Two things live here, and nothing else:
  * `frontend`  -- the stereo ORB front-end (FAST -> rBRIEF -> brute-force Hamming -> get_landmarks)
                   that replaces the external image-processor nodelet of the reference
                   (launch/stereo.launch:33-55; consumed at batch.py:149-154);
  * `gtsam`     -- a gtsam-shaped module (Values, NonlinearFactorGraph, GenericStereoFactor3D,
                   LevenbergMarquardtOptimizer, ...) exposing exactly the API batch.py uses
                   (batch.py:19-26, 270-305, 337), whose optimize() runs on hand-written HIP kernels.

All arithmetic on the path runs in libvus_hip.so (csrc/, C ABI in include/vus.h).  There is no CPU
fallback: importing `_lib` without the built library, or calling a kernel without a GPU, raises.
"""
__version__ = "0.1.0"
