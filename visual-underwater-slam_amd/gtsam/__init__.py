"""A gtsam-shaped module: the drop-in boundary of the BA hot path.

`/root/reference/batch.py` builds its factor graph and solves it exclusively through the Python
`gtsam` API (import list batch.py:19-26; graph build :270-305; solve :337; read-back :57-68).  This
module exposes the same names with the same argument meaning, so that

    import visual_underwater_slam_amd.gtsam as gtsam
    from visual_underwater_slam_amd.gtsam.symbol_shorthand import B, V, X, L

is the only change batch.py needs for the stereo path; `LevenbergMarquardtOptimizer.optimize()` then
runs on the MI355X kernels (ba.py / csrc/ba.hip).

Scope: GenericStereoFactor3D, PriorFactorPose3, PriorFactorVector, ImuFactor (with
PreintegratedImuMeasurements) and the DVL velocity factor (DvlVelocityFactor, the well-formed
replacement of the reference's CustomFactor) are solved on the GPU.  A generic gtsam.CustomFactor
(arbitrary Python callback) can be constructed and added, but optimize() refuses it, loudly.

Host-side classes here hold plain numpy values; no BA arithmetic happens in Python.
Errors surface as RuntimeError, as pybind11 does for gtsam's C++ exceptions.
"""
import math
from array import array
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from . import symbol_shorthand  # noqa: F401
from . import utils  # noqa: F401
from .symbol_shorthand import symbol, symbolChr, symbolIndex  # noqa: F401

__all__ = [
    "Point3", "Rot3", "Pose3", "Cal3_S2Stereo", "Cal3_S2", "StereoPoint2", "noiseModel", "imuBias",
    "GenericStereoFactor3D", "PriorFactorPose3", "PriorFactorVector", "PriorFactorPoint3",
    "PriorFactorConstantBias", "BetweenFactorConstantBias", "ImuFactor", "CustomFactor", "DvlVelocityFactor",
    "PreintegrationParams", "PreintegratedImuMeasurements", "NavState", "ISAM2", "ConstantTwistScenario",
    "PinholeCameraCal3_S2",
    "NonlinearFactorGraph", "Values", "LevenbergMarquardtParams", "LevenbergMarquardtOptimizer",
    "StereoFactorBlock", "symbol_shorthand", "symbol",
]


# ---------------------------------------------------------------------------------------------
# geometry
def Point3(x=0.0, y=0.0, z=0.0):
    """gtsam.Point3 is a numpy 3-vector in the Python wrapper (batch.py:46,84,132,166)."""
    if isinstance(x, (list, tuple, np.ndarray)):
        return np.asarray(x, dtype=float).reshape(3).copy()
    return np.array([x, y, z], dtype=float)


def _skew(w):
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


class Rot3:
    def __init__(self, R=None):
        self._R = np.eye(3) if R is None else np.asarray(R, dtype=float).reshape(3, 3).copy()

    @staticmethod
    def Quaternion(w, x, y, z):                    # batch.py:47,131  (w first)
        n = math.sqrt(w * w + x * x + y * y + z * z)
        w, x, y, z = w / n, x / n, y / n, z / n
        return Rot3([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])

    @staticmethod
    def Rodrigues(wx, wy=None, wz=None):           # batch.py:190
        w = np.asarray(wx, dtype=float).reshape(3) if wy is None else np.array([wx, wy, wz], dtype=float)
        return Rot3.Expmap(w)

    @staticmethod
    def Expmap(w):
        w = np.asarray(w, dtype=float).reshape(3)
        th2 = float(w @ w)
        K = _skew(w)
        if th2 <= np.finfo(float).eps:
            return Rot3(np.eye(3) + K)
        th = math.sqrt(th2)
        return Rot3(np.eye(3) + math.sin(th) / th * K + 2.0 * math.sin(0.5 * th) ** 2 / th2 * (K @ K))

    @staticmethod
    def Rx(t):
        c, s = math.cos(t), math.sin(t)
        return Rot3([[1, 0, 0], [0, c, -s], [0, s, c]])

    @staticmethod
    def Ry(t):
        c, s = math.cos(t), math.sin(t)
        return Rot3([[c, 0, s], [0, 1, 0], [-s, 0, c]])

    @staticmethod
    def Rz(t):
        c, s = math.cos(t), math.sin(t)
        return Rot3([[c, -s, 0], [s, c, 0], [0, 0, 1]])

    @staticmethod
    def Ypr(y, p, r):
        return Rot3(Rot3.Rz(y)._R @ Rot3.Ry(p)._R @ Rot3.Rx(r)._R)

    def matrix(self):
        return self._R.copy()

    def inverse(self):
        return Rot3(self._R.T)

    def compose(self, other):
        return Rot3(self._R @ other._R)

    def rotate(self, p):
        return self._R @ np.asarray(p, dtype=float).reshape(3)

    def __mul__(self, other):
        return self.compose(other)

    def equals(self, other, tol=1e-9):
        return bool(np.allclose(self._R, other._R, atol=tol))

    def __repr__(self):
        return f"Rot3(\n{self._R}\n)"


class Pose3:
    """Camera/body pose: rotation R and translation t (body-to-world), gtsam.Pose3."""

    def __init__(self, r=None, t=None):
        if r is None:
            self._R, self._t = np.eye(3), np.zeros(3)
        elif isinstance(r, Pose3):
            self._R, self._t = r._R.copy(), r._t.copy()
        elif isinstance(r, Rot3):
            self._R = r.matrix()
            self._t = np.zeros(3) if t is None else np.asarray(t, dtype=float).reshape(3).copy()
        else:
            M = np.asarray(r, dtype=float)
            if M.shape != (4, 4):
                raise RuntimeError("Pose3(): expected (Rot3, Point3), a 4x4 matrix, or nothing")
            self._R, self._t = M[:3, :3].copy(), M[:3, 3].copy()

    # accessors used by batch.py:62,134-135,213
    def x(self):
        return float(self._t[0])

    def y(self):
        return float(self._t[1])

    def z(self):
        return float(self._t[2])

    def rotation(self):
        return Rot3(self._R)

    def translation(self):
        return self._t.copy()

    def matrix(self):
        M = np.eye(4)
        M[:3, :3], M[:3, 3] = self._R, self._t
        return M

    def inverse(self):
        return Pose3(Rot3(self._R.T), -self._R.T @ self._t)

    def compose(self, other):
        return Pose3(Rot3(self._R @ other._R), self._t + self._R @ other._t)

    def __mul__(self, other):
        return self.compose(other)

    def between(self, other):
        return self.inverse().compose(other)

    def transformFrom(self, p):
        return self._R @ np.asarray(p, dtype=float).reshape(3) + self._t

    def transformTo(self, p):
        return self._R.T @ (np.asarray(p, dtype=float).reshape(3) - self._t)

    def equals(self, other, tol=1e-9):
        return bool(np.allclose(self._R, other._R, atol=tol) and np.allclose(self._t, other._t, atol=tol))

    def flat12(self):
        """Row-major R followed by t: the device layout of include/vus.h."""
        return np.concatenate([self._R.reshape(-1), self._t])

    @staticmethod
    def from_flat12(v):
        v = np.asarray(v, dtype=float).reshape(12)
        return Pose3(Rot3(v[:9].reshape(3, 3)), v[9:])

    def __repr__(self):
        return f"Pose3(R=\n{self._R},\n t={self._t})"


class Cal3_S2Stereo:
    def __init__(self, fx=1.0, fy=1.0, s=0.0, u0=0.0, v0=0.0, b=1.0):   # batch.py:115
        self._v = (float(fx), float(fy), float(s), float(u0), float(v0), float(b))

    def fx(self):
        return self._v[0]

    def fy(self):
        return self._v[1]

    def skew(self):
        return self._v[2]

    def px(self):
        return self._v[3]

    def py(self):
        return self._v[4]

    def baseline(self):
        return self._v[5]

    def vector6(self):
        return np.array(self._v)

    def equals(self, other, tol=1e-9):
        return bool(np.allclose(self._v, other._v, atol=tol))


class Cal3_S2:
    def __init__(self, fx=1.0, fy=1.0, s=0.0, u0=0.0, v0=0.0):
        self._v = (float(fx), float(fy), float(s), float(u0), float(v0))


class StereoPoint2:
    def __init__(self, uL=0.0, uR=0.0, v=0.0):                           # batch.py:301
        self._m = np.array([uL, uR, v], dtype=float)

    def uL(self):
        return float(self._m[0])

    def uR(self):
        return float(self._m[1])

    def v(self):
        return float(self._m[2])

    def vector(self):
        return self._m.copy()


# ---------------------------------------------------------------------------------------------
# noise models (batch.py:95-98,118,189)
class _NoiseModel:
    def __init__(self, sigmas):
        self._sigmas = np.asarray(sigmas, dtype=float).reshape(-1).copy()
        if (self._sigmas <= 0).any():
            raise RuntimeError("noise model sigmas must be positive")

    def sigmas(self):
        return self._sigmas.copy()

    def dim(self):
        return int(self._sigmas.size)

    def is_isotropic(self):
        return bool(np.all(self._sigmas == self._sigmas[0]))


class _Diagonal(_NoiseModel):
    @staticmethod
    def Sigmas(sigmas):
        return _Diagonal(sigmas)

    @staticmethod
    def Variances(v):
        return _Diagonal(np.sqrt(np.asarray(v, dtype=float)))

    @staticmethod
    def Precisions(p):
        return _Diagonal(1.0 / np.sqrt(np.asarray(p, dtype=float)))


class _Isotropic(_NoiseModel):
    @staticmethod
    def Sigma(dim, sigma):
        return _Isotropic(np.full(int(dim), float(sigma)))

    @staticmethod
    def Variance(dim, variance):
        return _Isotropic(np.full(int(dim), math.sqrt(float(variance))))


class _Unit(_NoiseModel):
    @staticmethod
    def Create(dim):
        return _Unit(np.ones(int(dim)))


class noiseModel:  # namespace, as in gtsam
    Base = _NoiseModel
    Diagonal = _Diagonal
    Isotropic = _Isotropic
    Unit = _Unit


class _ConstantBias:
    def __init__(self, biasAcc=None, biasGyro=None):
        self._a = np.zeros(3) if biasAcc is None else np.asarray(biasAcc, dtype=float).reshape(3)
        self._g = np.zeros(3) if biasGyro is None else np.asarray(biasGyro, dtype=float).reshape(3)

    def vector(self):
        return np.concatenate([self._a, self._g])

    def accelerometer(self):
        return self._a.copy()

    def gyroscope(self):
        return self._g.copy()


class imuBias:  # namespace (batch.py:92)
    ConstantBias = _ConstantBias


# ---------------------------------------------------------------------------------------------
# factors
class _Factor:
    def __init__(self, keys):
        self._keys = [int(k) for k in keys]

    def keys(self):
        return list(self._keys)

    def size(self):
        return len(self._keys)


class GenericStereoFactor3D(_Factor):
    """GenericStereoFactor<Pose3, Point3>(measured, model, poseKey, landmarkKey, K)  (batch.py:300-304)."""

    def __init__(self, measured: StereoPoint2, model: _NoiseModel, poseKey: int, landmarkKey: int,
                 K: Cal3_S2Stereo):
        super().__init__([poseKey, landmarkKey])
        if model.dim() != 3:
            raise RuntimeError("GenericStereoFactor3D needs a 3-dimensional noise model")
        self._measured, self._model, self._K = measured, model, K

    def measured(self):
        return self._measured

    def calibration(self):
        return self._K

    def noiseModel(self):
        return self._model


class StereoFactorBlock(_Factor):
    """EXTENSION (not in gtsam): many GenericStereoFactor3D factors sharing one noise model and one
    calibration, held as arrays.  It is the vectorised form of the emission loop batch.py:296-305 for
    graphs with millions of observations, where one Python object per factor is the bottleneck."""

    def __init__(self, measured, model: _NoiseModel, poseKeys, landmarkKeys, K: Cal3_S2Stereo):
        self.meas = np.ascontiguousarray(measured, dtype=float).reshape(-1, 3)
        self.pose_keys = np.ascontiguousarray(poseKeys, dtype=np.int64).reshape(-1)
        self.landmark_keys = np.ascontiguousarray(landmarkKeys, dtype=np.int64).reshape(-1)
        if not (len(self.meas) == len(self.pose_keys) == len(self.landmark_keys)):
            raise RuntimeError("StereoFactorBlock: arrays differ in length")
        if model.dim() != 3:
            raise RuntimeError("StereoFactorBlock needs a 3-dimensional noise model")
        self._model, self._K = model, K
        self._keys = None

    def keys(self):
        return np.unique(np.concatenate([self.pose_keys, self.landmark_keys])).tolist()

    def size(self):
        return len(self.meas)

    def __len__(self):
        return len(self.meas)


class _PriorFactor(_Factor):
    def __init__(self, key, prior, model):
        super().__init__([key])
        self._prior, self._model = prior, model

    def prior(self):
        return self._prior

    def noiseModel(self):
        return self._model


class PriorFactorPose3(_PriorFactor):
    def __init__(self, key, prior: Pose3, model: _NoiseModel):           # batch.py:281
        if model.dim() != 6:
            raise RuntimeError("PriorFactorPose3 needs a 6-dimensional noise model")
        super().__init__(key, Pose3(prior), model)


class PriorFactorVector(_PriorFactor):
    def __init__(self, key, prior, model: _NoiseModel):                  # batch.py:282
        prior = np.asarray(prior, dtype=float).reshape(-1).copy()
        if model.dim() != prior.size:
            raise RuntimeError("PriorFactorVector: noise model dimension differs from the vector's")
        super().__init__(key, prior, model)


class PriorFactorPoint3(PriorFactorVector):
    pass


class PriorFactorConstantBias(_PriorFactor):
    def __init__(self, key, prior: _ConstantBias, model: _NoiseModel):
        super().__init__(key, prior, model)


class BetweenFactorConstantBias(_Factor):
    def __init__(self, key1, key2, measured, model):
        super().__init__([key1, key2])
        self._measured, self._model = measured, model


class PreintegrationParams:
    """Holds the IMU noise parameters batch.py sets at :181-187 (consumed by gtsam/imu.py's Preintegrator when a PreintegratedImuMeasurements is built from them)."""

    def __init__(self, n_gravity):
        self.n_gravity = np.asarray(n_gravity, dtype=float)
        self.accelerometerCovariance = np.eye(3)
        self.gyroscopeCovariance = np.eye(3)
        self.integrationCovariance = np.eye(3)
        self.use2ndOrderCoriolis = False
        self.omegaCoriolis = np.zeros(3)

    @staticmethod
    def MakeSharedU(g=9.81):
        return PreintegrationParams([0.0, 0.0, -float(g)])

    @staticmethod
    def MakeSharedD(g=9.81):
        return PreintegrationParams([0.0, 0.0, float(g)])

    def setAccelerometerCovariance(self, c):
        self.accelerometerCovariance = np.asarray(c, dtype=float)

    def setGyroscopeCovariance(self, c):
        self.gyroscopeCovariance = np.asarray(c, dtype=float)

    def setIntegrationCovariance(self, c):
        self.integrationCovariance = np.asarray(c, dtype=float)

    def setUse2ndOrderCoriolis(self, f):
        self.use2ndOrderCoriolis = bool(f)

    def setOmegaCoriolis(self, w):
        self.omegaCoriolis = np.asarray(w, dtype=float)


class PreintegratedImuMeasurements:
    """gtsam.PreintegratedImuMeasurements(params, bias): on-manifold preintegration of the raw samples
    (batch.py:91,290,293), done on the host at graph-build time exactly like GTSAM does (imu.py)."""

    def __init__(self, params: PreintegrationParams, bias: Optional[_ConstantBias] = None):
        from .imu import Preintegrator
        self.params, self.bias = params, bias or _ConstantBias()
        self._pre = Preintegrator(self.bias.vector(), params.accelerometerCovariance, params.gyroscopeCovariance,
                                  params.integrationCovariance)

    def integrateMeasurement(self, acc, gyro, dt):
        self._pre.integrate(acc, gyro, dt)

    def resetIntegration(self):
        self._pre.reset()

    def deltaTij(self):
        return float(self._pre.dt)

    def deltaRij(self):
        return Rot3(self._pre.dR)

    def deltaPij(self):
        return self._pre.dP.copy()

    def deltaVij(self):
        return self._pre.dV.copy()

    def preintMeasCov(self):
        return self._pre.cov.copy()


class ImuFactor(_Factor):
    """ImuFactor(pose_i, vel_i, pose_j, vel_j, bias, pim)  (batch.py:238).  Like gtsam, the factor copies the
    preintegrated measurement at construction, so resetIntegration() right after (batch.py:293) is safe."""

    def __init__(self, pose_i, vel_i, pose_j, vel_j, bias, pim: PreintegratedImuMeasurements):
        super().__init__([pose_i, vel_i, pose_j, vel_j, bias])
        if pim._pre.dt <= 0.0:
            raise RuntimeError("ImuFactor: the preintegrated interval is empty (no IMU sample between the keyframes)")
        if pim.params.use2ndOrderCoriolis or np.any(pim.params.omegaCoriolis != 0):
            raise NotImplementedError("Coriolis terms are not implemented (batch.py:186-187 disables them)")
        self.pim = pim._pre.packed()
        self.W = pim._pre.whitening().reshape(-1)
        self.gravity = np.asarray(pim.params.n_gravity, dtype=float).copy()


class DvlVelocityFactor(_Factor):
    """EXTENSION replacing the reference's DVL gtsam.CustomFactor (batch.py:196-250): same residual
    e = R_i * m - v_i on keys (V(i), X(i)), with the correct Jacobians (the callback of the reference returns
    3x3 Jacobians for a 6-dof Pose3 key, which no solver can use: SURVEY.md D7).
    DvlVelocityFactor(noiseModel, V(i), X(i), measured_body_velocity)."""

    def __init__(self, model: _NoiseModel, velKey: int, poseKey: int, measured):
        super().__init__([velKey, poseKey])
        if model.dim() != 3 or not model.is_isotropic():
            raise NotImplementedError("DvlVelocityFactor needs an isotropic 3-dimensional noise model (batch.py:98)")
        self._model = model
        self.measured = np.asarray(measured, dtype=float).reshape(3).copy()


class CustomFactor(_Factor):
    def __init__(self, model: _NoiseModel, keys: Sequence[int], error_function: Callable):       # batch.py:245-249
        super().__init__(keys)
        self._model, self._fn = model, error_function


class NavState:
    def __init__(self, pose=None, velocity=None):
        self._pose, self._v = pose or Pose3(), np.zeros(3) if velocity is None else np.asarray(velocity, float)


class ConstantTwistScenario:
    """Imported by batch.py:19-25 and never used; only the name has to exist for the import line."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError("ConstantTwistScenario is imported but unused by batch.py; not part of the hot path")


class PinholeCameraCal3_S2:
    """Imported by batch.py:19-25 and never used; only the name has to exist for the import line."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError("PinholeCameraCal3_S2 is imported but unused by batch.py; not part of the hot path")


class ISAM2:
    """Constructed but never used by batch.py (batch.py:79); iSAM2 is out of scope."""

    def __init__(self, *args, **kwargs):
        pass

    def update(self, *args, **kwargs):
        raise NotImplementedError("ISAM2 is outside the hot path (isam.py is declared broken: README.md:40-41)")


# ---------------------------------------------------------------------------------------------
# containers
class _ValueBlock:
    """Array-backed run of same-typed variables (EXTENSION, see Values.insert_point3_block): keys ascending,
    data [n,3] (Point3) or [n,12] (Pose3 as row-major R then t)."""
    __slots__ = ("kind", "keys", "data")

    def __init__(self, kind, keys, data):
        self.kind, self.keys, self.data = kind, keys, data

    def find(self, keys):
        """(position, found) of every key of the int64 array `keys` in this block."""
        if len(self.keys) == 0:
            return np.zeros(len(keys), np.int64), np.zeros(len(keys), bool)
        pos = np.minimum(np.searchsorted(self.keys, keys), len(self.keys) - 1)
        return pos, self.keys[pos] == keys


class _Slot:
    """Dictionary entry of a variable whose value lives in one of the Values' growing columnar stores."""
    __slots__ = ("kind", "row")

    def __init__(self, kind, row):
        self.kind, self.row = kind, row


_WIDTH = {"point3": 3, "pose3": 12}


class _Column:
    """Growing columnar store of same-typed variables inserted ONE BY ONE (the way batch.py:283-298 inserts them):
    keys and values are appended to flat C arrays at insertion time, so that packing a graph for the GPU, and writing
    its result back, never loops over the variables in Python.  A sorted view (keys ascending -> row) is built lazily
    and dropped by every insertion / erasure."""
    __slots__ = ("kind", "width", "keys", "data", "dead", "_view")

    def __init__(self, kind, other: Optional["_Column"] = None):
        self.kind, self.width = kind, _WIDTH[kind]
        self.keys = array("q", other.keys) if other is not None else array("q")
        self.data = array("d", other.data) if other is not None else array("d")
        self.dead = other.dead if other is not None else 0
        self._view = None

    def append(self, key, flat):
        self.keys.append(key)
        self.data.extend(flat)
        self._view = None
        return len(self.keys) - 1

    def row(self, r):
        w = self.width
        return np.array(self.data[w * r:w * (r + 1)])

    def set_row(self, r, flat):
        w = self.width
        self.data[w * r:w * (r + 1)] = array("d", flat)

    def kill(self, r):
        self.keys[r] = -1              # keys are symbols (chr << 56 | index) or plain non-negative integers
        self.dead += 1
        self._view = None

    def arrays(self):
        """(keys int64 [n], data f64 [n, width]) VIEWS of the live storage.  While such a view is alive the stores cannot
        grow (array.append raises BufferError), so no view ever leaves this class: callers get copies (take) and
        write through put."""
        k = np.frombuffer(self.keys, dtype=np.int64) if len(self.keys) else np.zeros(0, np.int64)
        d = (np.frombuffer(self.data, dtype=np.float64) if len(self.data) else np.zeros(0)).reshape(-1, self.width)
        return k, d

    def take(self, rows):
        """Copy of the given rows [m, width]."""
        return self.arrays()[1][rows]

    def put(self, rows, values):
        self.arrays()[1][rows] = values

    def view(self):
        """(sorted live keys, their rows)."""
        if self._view is None:
            k, _ = self.arrays()
            order = np.argsort(k, kind="stable")
            if self.dead:
                order = order[k[order] >= 0]
            self._view = (k[order], order)
        return self._view

    def find(self, keys):
        sk, rows = self.view()
        if len(sk) == 0:
            return np.zeros(len(keys), np.int64), np.zeros(len(keys), bool)
        pos = np.minimum(np.searchsorted(sk, keys), len(sk) - 1)
        return rows[pos], sk[pos] == keys

    def __len__(self):
        return len(self.keys) - self.dead


class Values:
    """gtsam.Values.  Storage is columnar from the moment of insertion: 3-vectors (Point3 landmarks, velocities) and
    Pose3 values inserted one by one (batch.py:283-298) are appended to growing flat arrays (`_Column`), whole runs
    inserted through the bulk EXTENSION methods are kept as arrays (`_ValueBlock`); only values of other types
    (imuBias.ConstantBias, vectors of other sizes, Rot3) are held as objects.  Packing a 50 000-landmark graph for the
    GPU and reading its result back therefore involve no per-variable Python work."""

    def __init__(self, other: Optional["Values"] = None):
        self._d: Dict[int, object] = dict(other._d) if other is not None else {}
        self._blk: List[_ValueBlock] = ([_ValueBlock(b.kind, b.keys, b.data.copy()) for b in other._blk]
                                        if other is not None else [])
        self._col = {k: _Column(k, other._col[k] if other is not None else None) for k in _WIDTH}

    # -- block helpers ------------------------------------------------------------------------------
    def _find_block(self, key):
        for b in self._blk:
            n = len(b.keys)
            if n:
                i = int(np.searchsorted(b.keys, key))
                if i < n and b.keys[i] == key:
                    return b, i
        return None, -1

    def _insert_block(self, kind, keys, data, width):
        keys = np.ascontiguousarray(keys, dtype=np.int64).reshape(-1)
        data = np.array(data, dtype=float).reshape(-1, width)
        if len(keys) != len(data):
            raise RuntimeError("Values: keys and values differ in length")
        order = np.argsort(keys, kind="stable")
        keys, data = keys[order], data[order]
        dup = len(keys) > 1 and bool((keys[1:] == keys[:-1]).any())
        if not dup and self._d:
            dup = any(int(k) in self._d for k in keys.tolist()) if len(keys) < len(self._d) else \
                bool(np.isin(np.fromiter(self._d.keys(), np.int64, len(self._d)), keys).any())
        for b in self._blk:
            dup = dup or bool(b.find(keys)[1].any())
        if dup:
            raise RuntimeError("Attempting to add a key-value pair with a key which already exists in the Values.")
        self._blk.append(_ValueBlock(kind, keys, data))

    def insert_point3_block(self, keys, points):
        """EXTENSION: vectorised `for k, p in zip(keys, points): insert(k, p)` (batch.py:297-298)."""
        self._insert_block("point3", keys, points, 3)

    def insert_pose3_block(self, keys, flat12):
        """EXTENSION: bulk insertion of Pose3 values given as [n,12] rows (row-major R, then t)."""
        self._insert_block("pose3", keys, flat12, 12)

    insert_points = insert_point3_block

    def _stores(self, kind):
        """Every array-backed store that can hold variables of `kind`: (find(keys) -> (rows, hit), take(rows) -> copy,
        put(rows, values), kind).  No view of a growing column is handed out (see _Column.arrays)."""
        for b in self._blk:
            yield b.find, b.data.__getitem__, b.data.__setitem__, b.kind
        for c in self._col.values():
            if len(c):
                yield c.find, c.take, c.put, c.kind

    def _rows(self, kind, keys, what):
        """[n,w] array of the variables `keys` (int64 array) of the given kind; raises like gtsam's at*()."""
        keys = np.ascontiguousarray(keys, dtype=np.int64).reshape(-1)
        out = np.empty((len(keys), _WIDTH[kind]))
        todo = np.ones(len(keys), bool)
        for find, take, _, k in self._stores(kind):
            pos, hit = find(keys)
            hit &= todo
            if hit.any():
                if k != kind:
                    bad = int(keys[np.nonzero(hit)[0][0]])
                    raise RuntimeError(f"Values: key \"{symbol_shorthand.key_string(bad)}\" does not hold what {what} asks for")
                out[hit] = take(pos[hit])
                todo &= ~hit
        for i in np.nonzero(todo)[0].tolist():      # what is left is absent, or an object of another type
            k = int(keys[i])
            if k not in self._d:
                raise RuntimeError(f"Attempting to at the key \"{symbol_shorthand.key_string(k)}\", "
                                   "which does not exist in the Values.")
            raise RuntimeError(f"Values: key \"{symbol_shorthand.key_string(k)}\" is not a "
                               f"{'Pose3' if kind == 'pose3' else 'Point3'}")
        return out

    def point3_block(self, keys):
        """EXTENSION: [n,3] array of the Point3 variables `keys` (bulk atPoint3)."""
        return self._rows("point3", keys, "atPoint3")

    def pose3_block(self, keys):
        """EXTENSION: [n,12] array (row-major R, then t) of the Pose3 variables `keys` (bulk atPose3)."""
        return self._rows("pose3", keys, "atPose3")

    def _pose3_table(self):
        """(sorted keys, [n,12]) of every Pose3 variable."""
        sk, rows = self._col["pose3"].view()
        ks, data = [sk], [self._col["pose3"].take(rows)]
        for b in self._blk:
            if b.kind == "pose3":
                ks.append(b.keys); data.append(b.data)
        keys, data = np.concatenate(ks), np.concatenate(data)
        order = np.argsort(keys, kind="stable")
        return keys[order], data[order]

    def _store_rows(self, kind, keys, rows):
        """Overwrite existing variables of `kind` (result write-back of the optimizer), vectorised."""
        keys = np.ascontiguousarray(keys, dtype=np.int64).reshape(-1)
        todo = np.ones(len(keys), bool)
        for find, _, put, k in self._stores(kind):
            if k != kind:
                continue
            pos, hit = find(keys)
            hit &= todo
            put(pos[hit], rows[hit])
            todo &= ~hit
        if todo.any():
            k = int(keys[np.nonzero(todo)[0][0]])
            raise RuntimeError(f"Values: key \"{symbol_shorthand.key_string(k)}\" is not a stored "
                               f"{'Pose3' if kind == 'pose3' else 'Point3'}")

    # -- gtsam API ---------------------------------------------------------------------------------
    def _put(self, key, v):
        if isinstance(v, Pose3):
            self._d[key] = _Slot("pose3", self._col["pose3"].append(key, v.flat12()))
        elif isinstance(v, np.ndarray) and v.size == 3:
            self._d[key] = _Slot("point3", self._col["point3"].append(key, v))
        else:
            self._d[key] = v

    def insert(self, key, value):                                        # batch.py:274,283-288,298
        key = int(key)
        if key in self._d or (self._blk and self._find_block(key)[0] is not None):
            raise RuntimeError(f"Attempting to add a key-value pair with key \"{symbol_shorthand.key_string(key)}\", "
                               "which already exists in the Values.")
        self._put(key, self._coerce(value))

    def update(self, key, value):
        key = int(key)
        v = self._coerce(value)
        b, i = self._find_block(key)
        held = b.kind if b is not None else None
        if b is None:
            if key not in self._d:
                raise RuntimeError(f"Requested to update a key-value pair with key \"{symbol_shorthand.key_string(key)}\", "
                                   "which does not exist in the Values.")
            cur = self._d[key]
            if not isinstance(cur, _Slot):       # an object-held value: replaced by whatever comes, as before
                del self._d[key]
                self._put(key, v)
                return
            held = cur.kind
        # gtsam refuses to change a variable's type through update(): same RuntimeError as _at() / _rows()
        ok = isinstance(v, Pose3) if held == "pose3" else (isinstance(v, np.ndarray) and v.size == 3)
        if not ok:
            raise RuntimeError(f"Values: key \"{symbol_shorthand.key_string(key)}\" holds a "
                               f"{'Pose3' if held == 'pose3' else 'Point3'}, not the {type(v).__name__} update() was given")
        flat = v.flat12() if held == "pose3" else v
        if b is not None:
            b.data[i] = flat
        else:
            self._col[held].set_row(self._d[key].row, flat)

    def insert_or_assign(self, key, value):
        if self.exists(key):
            self.update(key, value)
        else:
            self.insert(key, value)

    @staticmethod
    def _coerce(value):
        if isinstance(value, (Pose3, Rot3, _ConstantBias)):
            return value
        return np.asarray(value, dtype=float).reshape(-1).copy()

    def exists(self, key):                                               # batch.py:60,297
        key = int(key)
        return key in self._d or (bool(self._blk) and self._find_block(key)[0] is not None)

    def erase(self, key):
        key = int(key)
        b, i = self._find_block(key)
        if b is not None:
            b.keys, b.data = np.delete(b.keys, i), np.delete(b.data, i, axis=0)
            return
        if key not in self._d:
            raise RuntimeError(f"key \"{symbol_shorthand.key_string(key)}\" does not exist in the Values")
        cur = self._d.pop(key)
        if isinstance(cur, _Slot):
            self._col[cur.kind].kill(cur.row)

    def _at(self, key, kind, name):
        key = int(key)
        if key in self._d:
            v = self._d[key]
            if isinstance(v, _Slot):
                r = self._col[v.kind].row(v.row)
                v = Pose3.from_flat12(r) if v.kind == "pose3" else r
        else:
            b, i = self._find_block(key)
            if b is None:
                raise RuntimeError(f"Attempting to {name} the key \"{symbol_shorthand.key_string(key)}\", "
                                   "which does not exist in the Values.")
            v = Pose3.from_flat12(b.data[i]) if b.kind == "pose3" else b.data[i].copy()
        if not isinstance(v, kind):
            raise RuntimeError(f"Values: key \"{symbol_shorthand.key_string(key)}\" holds a "
                               f"{type(v).__name__}, not what {name} asks for")
        return v

    def atPose3(self, key):                                              # batch.py:61,211
        return self._at(key, Pose3, "atPose3")

    def atVector(self, key):                                             # batch.py:210
        return self._at(key, np.ndarray, "atVector").copy()

    def atPoint3(self, key):
        v = self._at(key, np.ndarray, "atPoint3")
        if v.size != 3:
            raise RuntimeError("atPoint3: value is not a 3-vector")
        return v.copy()

    def atConstantBias(self, key):
        return self._at(key, _ConstantBias, "atConstantBias")

    def keys(self):
        ks = list(self._d)
        for b in self._blk:
            ks.extend(b.keys.tolist())
        return sorted(ks)

    def size(self):
        return len(self._d) + sum(len(b.keys) for b in self._blk)

    def __len__(self):
        return self.size()


class NonlinearFactorGraph:
    """gtsam.NonlinearFactorGraph.  GenericStereoFactor3D factors are ALSO recorded column-wise at the moment they are
    added (measurement, pose key, landmark key appended to flat arrays), so that the optimizer packs a graph of two
    million such objects (batch.py:300-305 pushes one per observation) without visiting them again."""

    def __init__(self):
        self._factors: List[_Factor] = []
        self._other: List[_Factor] = []          # everything that is not a single GenericStereoFactor3D
        self._st_meas, self._st_pk, self._st_lk = array("d"), array("q"), array("q")
        self._st_model = self._st_K = None       # the one noise model / calibration the stereo factors share ...
        self._st_mixed = False                   # ... or the fact that they do not (refused at optimize())

    def _record(self, factor):
        self._factors.append(factor)
        if type(factor) is GenericStereoFactor3D:
            m, K = factor._model, factor._K
            if self._st_model is None:
                self._st_model, self._st_K = m, K
            elif (m is not self._st_model and not np.array_equal(m._sigmas, self._st_model._sigmas)) or \
                    (K is not self._st_K and not K.equals(self._st_K)):
                self._st_mixed = True
            self._st_meas.extend(factor._measured._m)
            self._st_pk.append(factor._keys[0])
            self._st_lk.append(factor._keys[1])
        else:
            self._other.append(factor)

    def add(self, factor):                                               # batch.py:281-282
        self._record(factor)

    def push_back(self, factor):                                         # batch.py:291,292,305
        self._record(factor)

    def _stereo_columns(self):
        """(meas [n,3], pose keys [n], landmark keys [n], model, K, mixed) of the single stereo factors, in graph order."""
        n = len(self._st_pk)
        if n == 0:
            return np.zeros((0, 3)), np.zeros(0, np.int64), np.zeros(0, np.int64), None, None, False
        # copies (46 MB at two million factors, ~10 ms): a numpy VIEW of the growing arrays would pin their buffers and
        # make the next push_back fail with BufferError
        return (np.frombuffer(self._st_meas, dtype=np.float64).reshape(n, 3).copy(),
                np.frombuffer(self._st_pk, dtype=np.int64).copy(), np.frombuffer(self._st_lk, dtype=np.int64).copy(),
                self._st_model, self._st_K, self._st_mixed)

    def size(self):
        return len(self._factors)

    def nrFactors(self):
        return sum(f.size() if isinstance(f, StereoFactorBlock) else 1 for f in self._other) + len(self._st_pk)

    def at(self, i):
        return self._factors[i]

    def keys(self):
        out = set()
        for f in self._other:
            out.update(f.keys())
        out.update(self._st_pk.tolist())
        out.update(self._st_lk.tolist())
        return sorted(out)

    def error(self, values: Values) -> float:
        """0.5 * sum of squared whitened residuals, evaluated on the GPU."""
        from .optimizer import graph_error
        return graph_error(self, values)

    def saveGraph(self, path, values=None):                              # batch.py:338
        """Graphviz DOT of the factor graph (variables as circles, factors as dots)."""
        ks = self.keys()
        with open(path, "w") as f:
            f.write("graph {\n  size=\"5,5\";\n\n")
            for k in ks:
                f.write(f"  var{k}[label=\"{symbol_shorthand.key_string(k)}\"];\n")
            f.write("\n")
            n = 0
            for fac in self._factors:
                if isinstance(fac, StereoFactorBlock):
                    for pk, lk in zip(fac.pose_keys.tolist(), fac.landmark_keys.tolist()):
                        f.write(f"  factor{n}[label=\"\", shape=point];\n  var{pk}--factor{n};\n  var{lk}--factor{n};\n")
                        n += 1
                    continue
                f.write(f"  factor{n}[label=\"\", shape=point];\n")
                for k in fac.keys():
                    f.write(f"  var{k}--factor{n};\n")
                n += 1
            f.write("}\n")


from .optimizer import LevenbergMarquardtParams, LevenbergMarquardtOptimizer  # noqa: E402
