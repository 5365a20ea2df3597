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
    """State of one preintegration interval; integrate() consumes one (acc, gyro, dt) sample."""

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
