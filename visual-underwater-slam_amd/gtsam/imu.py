"""IMU preintegration on the host, as GTSAM does it inside PreintegratedImuMeasurements
(/root/reference/batch.py:90-92,178-193,289-293 use it through pybind).  Restated from Forster et al.,
"On-Manifold Preintegration for Real-Time Visual-Inertial Odometry" (TRO 2017); O(#IMU samples) work at
graph-build time, not on the hot path.  The packed record layout (148 doubles) is shared with the
kernels (include/vus.h, vus_nav_factors.imu_pim) and with the oracle (oracle/vus_oracle_nav.c PIM_*)."""
import numpy as np

PIM_DT, PIM_DR, PIM_DP, PIM_DV = 0, 1, 10, 13
PIM_DR_DBG, PIM_DP_DBA, PIM_DP_DBG, PIM_DV_DBA, PIM_DV_DBG = 16, 25, 34, 43, 52
PIM_BIAS, PIM_COV, PIM_DOUBLES = 61, 67, 148


def skew(w):
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


def so3_expmap(w):
    th2 = float(w @ w)
    W = skew(w)
    if th2 <= np.finfo(float).eps:
        return np.eye(3) + W
    th = np.sqrt(th2)
    return np.eye(3) + np.sin(th) / th * W + 2.0 * np.sin(0.5 * th) ** 2 / th2 * (W @ W)


def so3_jr(w):
    th2 = float(w @ w)
    W = skew(w)
    if th2 < 1e-10:
        a, b = 0.5 - th2 / 24.0, 1.0 / 6.0 - th2 / 120.0
    else:
        th = np.sqrt(th2)
        a, b = (1.0 - np.cos(th)) / th2, (th - np.sin(th)) / (th2 * th)
    return np.eye(3) - a * W + b * (W @ W)


class Preintegrator:
    """State of one preintegration interval; integrate() consumes one (acc, gyro, dt) sample.

    The samples are BUFFERED and integrated by the library's host function vus_imu_preintegrate (C++, like GTSAM's own
    integrateMeasurement) the next time the state is read: a keyframe interval of 40 samples is one native call instead
    of 40 rounds of small numpy products (78 of the 92 ms of the 50-keyframe end-to-end sequence).  ReferencePreintegrator
    below is the numpy restatement the native function is tested against (and through it against the oracle)."""

    def __init__(self, bias_hat, acc_cov, gyro_cov, int_cov):
        self.bias_hat = np.asarray(bias_hat, dtype=float).reshape(6).copy()
        self.acc_cov, self.gyro_cov, self.int_cov = (np.ascontiguousarray(np.asarray(c, dtype=float).reshape(3, 3))
                                                     for c in (acc_cov, gyro_cov, int_cov))
        self.reset()

    def reset(self):
        self._rec = np.zeros(PIM_DOUBLES)
        self._rec[PIM_DR:PIM_DR + 9] = np.eye(3).reshape(-1)
        self._rec[PIM_BIAS:PIM_BIAS + 6] = self.bias_hat
        self._pending = []

    def integrate(self, acc, gyro, dt):
        dt = float(dt)
        if not dt > 0.0:
            raise ValueError(f"integrate: dt={dt} is not positive")
        a, w = np.asarray(acc, dtype=float).reshape(3), np.asarray(gyro, dtype=float).reshape(3)
        self._pending.append((a[0], a[1], a[2], w[0], w[1], w[2], dt))

    def _flush(self, whiten=None):
        if not self._pending and whiten is None:
            return
        from .. import _lib
        smp = np.ascontiguousarray(np.array(self._pending, dtype=np.float64).reshape(-1, 7))
        _lib.call("vus_imu_preintegrate", self._rec.ctypes.data, smp.ctypes.data, len(smp), self.acc_cov.ctypes.data,
                  self.gyro_cov.ctypes.data, self.int_cov.ctypes.data, None if whiten is None else whiten.ctypes.data)
        self._pending = []

    def _mat(self, off):
        self._flush()
        return self._rec[off:off + 9].reshape(3, 3).copy()

    dt = property(lambda self: (self._flush(), float(self._rec[PIM_DT]))[1])
    dR = property(lambda self: self._mat(PIM_DR))
    dP = property(lambda self: (self._flush(), self._rec[PIM_DP:PIM_DP + 3].copy())[1])
    dV = property(lambda self: (self._flush(), self._rec[PIM_DV:PIM_DV + 3].copy())[1])
    dR_dbg = property(lambda self: self._mat(PIM_DR_DBG))
    dP_dba = property(lambda self: self._mat(PIM_DP_DBA))
    dP_dbg = property(lambda self: self._mat(PIM_DP_DBG))
    dV_dba = property(lambda self: self._mat(PIM_DV_DBA))
    dV_dbg = property(lambda self: self._mat(PIM_DV_DBG))
    cov = property(lambda self: (self._flush(), self._rec[PIM_COV:PIM_COV + 81].reshape(9, 9).copy())[1])

    def packed(self):
        self._flush()
        return self._rec.copy()

    def whitening(self):
        """W = L^-1 with cov = L L^T: |W r|^2 = r^T cov^-1 r (what gtsam's Gaussian noise model does)."""
        W = np.zeros((9, 9))
        self._flush(whiten=W)
        return W


class ReferencePreintegrator:
    """The same recursion in numpy, sample by sample (tests: the native function against this, this against the oracle)."""

    def __init__(self, bias_hat, acc_cov, gyro_cov, int_cov):
        self.bias_hat = np.asarray(bias_hat, dtype=float).reshape(6).copy()
        self.acc_cov, self.gyro_cov, self.int_cov = (np.asarray(c, dtype=float).reshape(3, 3) for c in (acc_cov, gyro_cov, int_cov))
        self.reset()

    def reset(self):
        self.dt = 0.0
        self.dR, self.dP, self.dV = np.eye(3), np.zeros(3), np.zeros(3)
        self.dR_dbg = np.zeros((3, 3))
        self.dP_dba, self.dP_dbg = np.zeros((3, 3)), np.zeros((3, 3))
        self.dV_dba, self.dV_dbg = np.zeros((3, 3)), np.zeros((3, 3))
        self.cov = np.zeros((9, 9))

    def integrate(self, acc, gyro, dt):
        dt = float(dt)
        a = np.asarray(acc, dtype=float).reshape(3) - self.bias_hat[:3]
        w = np.asarray(gyro, dtype=float).reshape(3) - self.bias_hat[3:]
        dRinc, Jr = so3_expmap(w * dt), so3_jr(w * dt)
        RaX = self.dR @ skew(a)
        A = np.eye(9)
        A[0:3, 0:3] = dRinc.T
        A[3:6, 0:3] = -0.5 * dt * dt * RaX
        A[3:6, 6:9] = np.eye(3) * dt
        A[6:9, 0:3] = -dt * RaX
        B = np.zeros((9, 3)); C = np.zeros((9, 3))
        B[3:6] = 0.5 * dt * dt * self.dR
        B[6:9] = dt * self.dR
        C[0:3] = dt * Jr
        cov = A @ self.cov @ A.T + B @ (self.acc_cov / dt) @ B.T + C @ (self.gyro_cov / dt) @ C.T
        cov[3:6, 3:6] += self.int_cov * dt
        self.cov = cov
        t = RaX @ self.dR_dbg
        self.dP_dba = self.dP_dba + self.dV_dba * dt - 0.5 * dt * dt * self.dR
        self.dP_dbg = self.dP_dbg + self.dV_dbg * dt - 0.5 * dt * dt * t
        self.dV_dba = self.dV_dba - dt * self.dR
        self.dV_dbg = self.dV_dbg - dt * t
        self.dR_dbg = dRinc.T @ self.dR_dbg - dt * Jr
        Ra = self.dR @ a
        self.dP = self.dP + self.dV * dt + 0.5 * dt * dt * Ra
        self.dV = self.dV + dt * Ra
        self.dR = self.dR @ dRinc
        self.dt += dt

    def packed(self):
        p = np.zeros(PIM_DOUBLES)
        p[PIM_DT] = self.dt
        p[PIM_DR:PIM_DR + 9] = self.dR.reshape(-1)
        p[PIM_DP:PIM_DP + 3], p[PIM_DV:PIM_DV + 3] = self.dP, self.dV
        for off, M in ((PIM_DR_DBG, self.dR_dbg), (PIM_DP_DBA, self.dP_dba), (PIM_DP_DBG, self.dP_dbg),
                       (PIM_DV_DBA, self.dV_dba), (PIM_DV_DBG, self.dV_dbg)):
            p[off:off + 9] = M.reshape(-1)
        p[PIM_BIAS:PIM_BIAS + 6] = self.bias_hat
        p[PIM_COV:PIM_COV + 81] = self.cov.reshape(-1)
        return p

    def whitening(self):
        """W = L^-1 with cov = L L^T: |W r|^2 = r^T cov^-1 r (what gtsam's Gaussian noise model does)."""
        L = np.linalg.cholesky(self.cov)
        return np.linalg.solve(L, np.eye(9))
