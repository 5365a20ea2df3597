"""Opportunistic cross-check against the engines the reference actually runs: OpenCV (the external
image-processor nodelet, /root/reference/launch/stereo.launch:33-55, README.md:18-21) and GTSAM
(/root/reference/batch.py:337).

TEST INFRASTRUCTURE ONLY (same rule as oracle.py).  Neither package is installed in the build container or
expected on the GPU box; nothing is fetched or installed: `importlib.util.find_spec` is a
local lookup.  When a package IS importable, the repo's own synthetic data is pushed through it and compared with
the oracle (tests/test_reference_engines.py) and it is timed as the CPU baseline (bench.py, kind "reference");
when it is absent every caller says so in plain words and falls back to the C port.  No file of the reference is
shipped, imported or executed.
"""
import importlib
import importlib.util

import numpy as np


def probe():
    """{'cv2': version or None, 'gtsam': version or None} -- local lookup only."""
    out = {}
    for name in ("cv2", "gtsam"):
        ver = None
        try:
            if importlib.util.find_spec(name) is not None:
                mod = importlib.import_module(name)
                ver = str(getattr(mod, "__version__", "unknown"))
        except Exception:        # a broken install counts as absent
            ver = None
        out[name] = ver
    return out


def describe(p=None):
    p = p or probe()
    return ", ".join(f"{k} {'absent' if v is None else v}" for k, v in p.items())


class EngineApiError(RuntimeError):
    """Driving the third-party API failed (wrong build, missing symbol): not a parity verdict."""


# ---------------------------------------------------------------------------------------------
# OpenCV
def cv2_fast_corner_set(img, thr=10):
    """Boolean map of FAST-9/16 corners (no non-max suppression) of one uint8 image, by OpenCV."""
    import cv2
    try:
        det = cv2.FastFeatureDetector_create(threshold=int(thr), nonmaxSuppression=False,
                                             type=cv2.FAST_FEATURE_DETECTOR_TYPE_9_16)
        kps = det.detect(np.ascontiguousarray(img), None)
    except Exception as e:      # pragma: no cover - only reachable where cv2 exists
        raise EngineApiError(f"cv2.FastFeatureDetector: {e}") from e
    m = np.zeros(img.shape, bool)
    for k in kps:
        m[int(round(k.pt[1])), int(round(k.pt[0]))] = True
    return m


def cv2_orb_detect_match(left, right, n_features=2000, thr=10):
    """cv2.ORB detect+describe on both images and BFMatcher(NORM_HAMMING) left->right: the reference pipeline's
    CPU cost for one stereo frame.  Returns the number of matches."""
    import cv2
    try:
        orb = cv2.ORB_create(nfeatures=int(n_features), fastThreshold=int(thr))
        _, dl = orb.detectAndCompute(left, None)
        _, dr = orb.detectAndCompute(right, None)
        if dl is None or dr is None:
            return 0
        return len(cv2.BFMatcher(cv2.NORM_HAMMING).match(dl, dr))
    except Exception as e:      # pragma: no cover
        raise EngineApiError(f"cv2.ORB / BFMatcher: {e}") from e


# ---------------------------------------------------------------------------------------------
# GTSAM
def gtsam_stereo_lm(seq, prior_on_gt=False):
    """The repo's synthetic stereo sequence (synth.ba_sequence dict) through the REAL gtsam with the call pattern of
    batch.py:270-305,337 (stereo factors + the X(0) prior).  Returns (poses [n,12] row-major R then t, points [m,3],
    final error, iterations)."""
    import gtsam
    from gtsam.symbol_shorthand import X, L
    try:
        n_kf = len(seq["poses_init"])
        K = gtsam.Cal3_S2Stereo(*[float(v) for v in seq["K"]])
        noise = gtsam.noiseModel.Isotropic.Sigma(3, float(seq["sigma"]))
        graph, initial = gtsam.NonlinearFactorGraph(), gtsam.Values()

        def pose(T):
            return gtsam.Pose3(gtsam.Rot3(np.asarray(T[:9], float).reshape(3, 3)), gtsam.Point3(*[float(v) for v in T[9:]]))
        T0 = seq["poses_gt"][0] if prior_on_gt else seq["poses_init"][0]
        graph.add(gtsam.PriorFactorPose3(X(0), pose(T0), gtsam.noiseModel.Diagonal.Sigmas(np.asarray(seq["prior_sigmas"], float))))
        for i in range(n_kf):
            initial.insert(X(i), pose(seq["poses_init"][i]))
        for j, p in enumerate(seq["points_init"]):
            initial.insert(L(j), gtsam.Point3(*[float(v) for v in p]))
        for a in range(len(seq["obs_pose"])):
            m = seq["meas"][a]
            graph.push_back(gtsam.GenericStereoFactor3D(gtsam.StereoPoint2(float(m[0]), float(m[1]), float(m[2])), noise,
                                                        X(int(seq["obs_pose"][a])), L(int(seq["obs_point"][a])), K))
        opt = gtsam.LevenbergMarquardtOptimizer(graph, initial, gtsam.LevenbergMarquardtParams())
        res = opt.optimize()
        poses = np.stack([np.concatenate([res.atPose3(X(i)).rotation().matrix().reshape(-1),
                                          np.asarray(res.atPose3(X(i)).translation(), float).reshape(-1)]) for i in range(n_kf)])
        points = np.stack([np.asarray(res.atPoint3(L(j)), float).reshape(-1) for j in range(len(seq["points_init"]))])
        return poses, points, float(opt.error()), int(opt.iterations())
    except Exception as e:      # pragma: no cover
        raise EngineApiError(f"gtsam: {e}") from e
