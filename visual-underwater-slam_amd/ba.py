"""Host side of the stereo bundle adjustment: device-resident problem, workspace, and the
Levenberg-Marquardt control loop that drives the HIP kernels of csrc/ba.hip through the C ABI.

Mirrors what `gtsam.LevenbergMarquardtOptimizer(graph, initial, params).optimize()` does at
/root/reference/batch.py:337 (GTSAM's iterate / tryLambda / checkConvergence logic with the default
LevenbergMarquardtParams), for graphs of GenericStereoFactor3D + PriorFactorPose3 factors.  All
arithmetic (residuals, Jacobians, Schur complement, band Cholesky, retraction, errors) runs on the
GPU; this file only sequences launches and reads back three scalars per lambda trial.
"""
import ctypes
import math
import time
from ctypes import c_double, c_int, c_void_p
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

from . import _lib, ba_pack


class _CProblem(ctypes.Structure):
    _fields_ = [("n_poses", c_int), ("n_points", c_int), ("n_obs", c_int), ("n_priors", c_int),
                ("K", c_void_p), ("inv_sigma", c_double), ("meas", c_void_p), ("obs_pose", c_void_p),
                ("obs_point", c_void_p), ("point_ptr", c_void_p), ("obs_ppos", c_void_p),
                ("pose_ptr", c_void_p), ("pobs_lidx", c_void_p), ("prior_pose", c_void_p),
                ("prior_T", c_void_p), ("prior_w", c_void_p), ("pose_stride", c_int)]


class _CTiles(ctypes.Structure):
    _fields_ = [("band", c_int), ("n_tiles", c_int), ("n_units", c_int), ("n_entries", c_int),
                ("unit_ptr", c_void_p), ("entries", c_void_p), ("order", c_void_p)]


@dataclass
class LMParams:
    """gtsam.LevenbergMarquardtParams() defaults."""
    lambdaInitial: float = 1e-5
    lambdaFactor: float = 10.0
    lambdaUpperBound: float = 1e5
    lambdaLowerBound: float = 0.0
    minModelFidelity: float = 1e-3
    maxIterations: int = 100
    relativeErrorTol: float = 1e-5
    absoluteErrorTol: float = 1e-5
    errorTol: float = 0.0
    diagonalDamping: bool = False
    useFixedLambdaFactor: bool = True


@dataclass
class LMReport:
    iterations: int = 0          # accepted steps (gtsam iterations())
    outer: int = 0               # linearisations
    tries: int = 0               # linear solves
    status: int = 1              # 0 converged, 1 max iterations, 2 lambda upper bound
    initial_error: float = 0.0
    final_error: float = 0.0
    final_lambda: float = 0.0
    err_hist: List[float] = field(default_factory=list)
    lambda_hist: List[float] = field(default_factory=list)
    seconds: float = 0.0
    setup_seconds: float = 0.0


def _i32(t):
    return t.to(torch.int32).contiguous()


def band_of(pk) -> int:
    """Widest keyframe span of a landmark = half-bandwidth, in pose blocks, of the reduced camera system."""
    band = pk.get("band")
    if band is None:
        if pk["n_obs"] == 0:
            return 0
        op, pptr = pk["obs_pose"], pk["point_ptr"].to(torch.int64)
        seen = pptr[1:] > pptr[:-1]
        first, last = op[pptr[:-1][seen]], op[pptr[1:][seen] - 1]
        band = int((last - first).max().item())
    return int(band)


def build_tiles_device(pk, band):
    """vus_ba_tiles of a packed problem (csrc/pack.hip: two launches and a radix sort): for every pair of 8-pose tiles
    within `band` poses of each other, the landmarks seen from both, in ascending order.  Returns a dict of device
    tensors + sizes; `band` >= the widest keyframe span of a landmark."""
    nP, nL, n_obs = pk["n_poses"], pk["n_points"], pk["n_obs"]
    dev = pk["obs_pose"].device
    n_tiles, dt1 = (nP + 7) // 8, (int(band) + 7) // 8 + 1
    i32 = dict(dtype=torch.int32, device=dev)
    out = {"band": int(band), "n_tiles": n_tiles, "n_units": n_tiles * dt1, "n_entries": 0,
           "unit_ptr": torch.zeros(n_tiles * dt1 + 1, **i32), "order": torch.arange(n_tiles * dt1, **i32),
           "entries": torch.zeros((1, 4), **i32)}
    if n_obs == 0 or nL == 0:
        return out
    p, st_ptr = _lib.ptr, _lib.current_stream_ptr()
    cp = _CProblem(nP, nL, n_obs, 0, None, 1.0, None, p(pk["obs_pose"]), p(pk["obs_point"]), p(pk["point_ptr"]),
                   p(pk["obs_ppos"]), p(pk["pose_ptr"]), p(pk["pobs_lidx"]), None, None, None, 1)
    cnt = torch.empty(nL, **i32)
    base = torch.empty(nL + 1, **i32)
    total = torch.empty(1, dtype=torch.int64, device=dev)
    _lib.call("vus_ba_tiles_count", ctypes.addressof(cp), p(cnt), st_ptr)
    _lib.call("vus_exclusive_scan_i32", p(cnt), nL, p(base), p(total), st_ptr)
    n = int(total.item())
    if n >= 2 ** 31:
        raise NotImplementedError(f"{n} tile-pair entries exceed the int32 entry index")
    out["n_entries"] = n
    out["entries"] = torch.empty((max(n, 1), 4), **i32)
    nbytes = int(_lib.load().vus_ba_tiles_work_bytes(n))
    work = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _lib.call("vus_ba_tiles_fill", ctypes.addressof(cp), int(band), p(base), n, p(out["unit_ptr"]), p(out["entries"]),
              p(out["order"]), p(work), nbytes, st_ptr)
    return out


class StereoBAProblem:
    """Packed, device-resident stereo BA problem (vus_ba_problem + vus_ba_tiles)."""

    def __init__(self, obs_pose, obs_point, meas, n_poses, n_points, K, sigma, prior_pose=None,
                 prior_T=None, prior_sigmas=None, device="cuda:0", band=None, pose_stride=1):
        _lib.require_gpu()
        _lib.load()
        dev = torch.device(device)
        t0 = time.perf_counter()

        def to_dev(x, dt):
            return torch.as_tensor(x).to(device=dev, dtype=dt).contiguous()
        if dev.type == "cuda":      # csrc/pack.hip: sorts and index arrays without torch's index operators
            pk = ba_pack.pack_observations_device(to_dev(obs_pose, torch.int32), to_dev(obs_point, torch.int32),
                                                  to_dev(meas, torch.float64), n_poses, n_points)
        else:
            pk = ba_pack.pack_observations(to_dev(obs_pose, torch.int64), to_dev(obs_point, torch.int64),
                                           to_dev(meas, torch.float64), n_poses, n_points)
        st = {"band": band_of(pk)}
        self.pk = pk
        self.device = dev
        self.n_poses, self.n_points, self.n_obs = int(n_poses), int(n_points), pk["n_obs"]
        self.pose_stride = int(pose_stride)
        self.n_nodes = self.pose_stride * int(n_poses)
        if self.pose_stride > 1:        # node layout with velocity nodes: pose blocks are 2 nodes apart,
            st["band"] = max(self.pose_stride * st["band"], 3)     # inertial factors reach 3 nodes back
        if band is not None:            # landmark-sharded solve: every rank allocates the global band
            if band < st["band"]:
                raise ValueError(f"band={band} is smaller than this problem's own band {st['band']}")
            st["band"] = int(band)
        self.band = st["band"]
        self.K = to_dev(K, torch.float64)
        assert self.K.numel() == 6
        self.sigma = float(sigma)
        if prior_pose is None or len(prior_pose) == 0:
            self.prior_pose = torch.zeros(0, dtype=torch.int32, device=dev)
            self.prior_T = torch.zeros((0, 12), dtype=torch.float64, device=dev)
            self.prior_w = torch.zeros((0, 6), dtype=torch.float64, device=dev)
        else:
            self.prior_pose = to_dev(prior_pose, torch.int32)
            self.prior_T = to_dev(prior_T, torch.float64).reshape(-1, 12)
            # reciprocal on the host: a torch elementwise kernel used once here costs the first call of a process ~20 ms
            # of lazily loaded code objects (tools/cold_phases.py)
            ps = prior_sigmas.detach().cpu().numpy() if torch.is_tensor(prior_sigmas) else prior_sigmas
            self.prior_w = to_dev(1.0 / np.asarray(ps, dtype=np.float64).reshape(-1, 6), torch.float64)
        n_pr = self.prior_pose.numel()
        p = _lib.ptr
        self.c_problem = _CProblem(self.n_poses, self.n_points, self.n_obs, n_pr, p(self.K), 1.0 / self.sigma,
                                   p(pk["meas"]), p(pk["obs_pose"]), p(pk["obs_point"]), p(pk["point_ptr"]),
                                   p(pk["obs_ppos"]), p(pk["pose_ptr"]), p(pk["pobs_lidx"]),
                                   p(self.prior_pose) if n_pr else None, p(self.prior_T) if n_pr else None,
                                   p(self.prior_w) if n_pr else None, int(pose_stride))
        # the tile pairs the Schur kernel walks (their band is in POSES), built once per graph on the device
        self.tiles = tl = build_tiles_device(pk, self.band // self.pose_stride)
        self.c_tiles = _CTiles(tl["band"], tl["n_tiles"], tl["n_units"], tl["n_entries"], p(tl["unit_ptr"]),
                               p(tl["entries"]), p(tl["order"]))
        torch.cuda.synchronize(dev)
        self.setup_seconds = time.perf_counter() - t0


class StereoBASolver:
    """Workspace + LM loop.  Buffers are allocated once; optimize() allocates nothing."""

    def __init__(self, problem: StereoBAProblem):
        self.P = problem
        dev, nP, nL, nO, B = problem.device, problem.n_poses, problem.n_points, problem.n_obs, problem.band
        nN = problem.n_nodes                       # camera-side nodes (= poses unless velocity nodes are interleaved)
        f64 = dict(dtype=torch.float64, device=dev)
        self.W = torch.empty((nO, 18), **f64)
        self.V = torch.empty((nL, 6), **f64)
        self.Vinv = torch.empty((nL, 6), **f64)
        self.gl = torch.empty((nL, 3), **f64)
        self.dl = torch.empty((nL, 3), **f64)
        self.Hpp = torch.empty((nP, 36), **f64)
        self.gp = torch.empty((nP, 6), **f64)
        self.gs = torch.empty((nN, 6), **f64)
        self.dp = torch.empty((nN, 6), **f64)
        self.Sband = torch.empty((nN, B + 1, 36), **f64)
        self.new_poses = torch.empty((nP, 12), **f64)
        self.new_points = torch.empty((nL, 3), **f64)
        self.work = torch.empty((2 * (nL + 1) + 8,), **f64)
        # one 40-byte record per lambda trial, read back with ONE device-to-host copy: [0] linearise error, [1] linearised
        # error at the step, [2] new error, [3] spare, [4] (as two int32) the band solve's status word
        self._trial = torch.zeros((5,), **f64)
        self.scal = self._trial[:4]
        self.status = self._trial[4:].view(torch.int32)[:1]
        # two-sided band solve (vus_ba_band_solve_split): worth it once the chain of panel steps is much longer than
        # the band; its workspace (pose-reversed copy of the lower half + the middle system) is allocated once
        self.band_rhs = 1
        self._alloc_band_work()
        self._unit_counter = torch.zeros(1, dtype=torch.int32, device=dev)      # the Schur kernel's unit queue

    SPLIT_MIN_EXTRA = 64       # use the two-sided solve when n_nodes >= 2 * band + this

    def _alloc_band_work(self):
        nN, B = self.P.n_nodes, self.P.band
        n = int(_lib.load().vus_ba_band_solve_work_doubles(nN, B, self.band_rhs))
        self.use_split = n > 0 and nN >= 2 * B + self.SPLIT_MIN_EXTRA
        self.band_work = torch.empty((n if self.use_split else 0,), dtype=torch.float64, device=self.P.device)

    # -- single kernels (also used by the parity tests) ------------------------------------------
    def _pp(self):
        return ctypes.addressof(self.P.c_problem)

    def error(self, poses, points) -> float:
        _lib.call("vus_ba_error", self._pp(), _lib.ptr(poses), _lib.ptr(points), _lib.ptr(self.scal),
                  _lib.ptr(self.work), _lib.current_stream_ptr())
        return float(self.scal[0].item())

    def linearize(self, poses, points):
        p = _lib.ptr
        _lib.call("vus_ba_linearize", self._pp(), p(poses), p(points), p(self.W), p(self.V), p(self.gl),
                  p(self.Hpp), p(self.gp), p(self.scal), p(self.work), _lib.current_stream_ptr())

    def schur(self, lam: float, Y=None):
        """Y: optional [n_obs,18] buffer that receives W Vinv (L-order); the kernel forms it on the fly and needs no
        such array (288 MB at configs[2])."""
        p = _lib.ptr
        _lib.call("vus_ba_schur", self._pp(), ctypes.addressof(self.P.c_tiles), float(lam), p(self.W), p(self.V),
                  p(self.gl), p(self.Hpp), p(self.gp), p(self.Vinv), p(Y), p(self.Sband), self.P.band, p(self.gs),
                  p(self._unit_counter), _lib.current_stream_ptr())

    def band_solve(self):
        p = _lib.ptr
        if self.use_split:
            _lib.call("vus_ba_band_solve_split", p(self.Sband), self.P.n_nodes, self.P.band, p(self.gs), p(self.dp),
                      p(self.status), p(self.band_work), _lib.current_stream_ptr())
        else:
            _lib.call("vus_ba_band_solve", p(self.Sband), self.P.n_nodes, self.P.band, p(self.gs), p(self.dp),
                      p(self.status), _lib.current_stream_ptr())

    def _window_expired(self, status) -> bool:
        """True if the trial should be redone: the persistent window kernel (band mode 3) gave up a bounded wait
        (VUS_STATUS_WINDOW_EXPIRED) -- its flag protocol needs every workgroup of its launch resident at once, and other
        work on the device (another rank or process on this GPU, a kernel of another stream holding CUs) can prevent
        that.  The launch-pair mode has no such demand: it is latched for the rest of the process (vus_ba_set_tuning is
        process-wide) and the caller redoes Schur + solve.  Every rank of a sharded solve sees the same status word and
        takes the same branch.  An expired wait under a mode the caller FORCED, or a second failure, is raised."""
        lib = _lib.load()
        if status != _lib.STATUS_WINDOW_EXPIRED or lib.vus_ba_get_tuning(_lib.TUNE_BAND_MODE) >= 0:
            return False
        import warnings
        warnings.warn("vus band solve: the persistent window kernel could not keep its workgroups resident (is other work "
                      "running on this GPU?); falling back to the launch-pair factorisation for the rest of the process")
        _lib.call("vus_ba_set_tuning", _lib.TUNE_BAND_MODE, 2)
        self.window_fallbacks = getattr(self, "window_fallbacks", 0) + 1
        return True

    def backsub(self):
        p = _lib.ptr
        _lib.call("vus_ba_backsub", self._pp(), p(self.W), p(self.Vinv), p(self.gl), p(self.dp), p(self.dl),
                  _lib.current_stream_ptr())

    def eval_step(self, poses, points):
        p = _lib.ptr
        _lib.call("vus_ba_eval_step", self._pp(), p(poses), p(points), p(self.dp), p(self.dl), p(self.new_poses),
                  p(self.new_points), p(self.scal[1:]), p(self.work), _lib.current_stream_ptr())

    # -- Levenberg-Marquardt ----------------------------------------------------------------------
    def optimize(self, poses: torch.Tensor, points: torch.Tensor, params: Optional[LMParams] = None,
                 aux=None):
        """poses [nP,12], points [nL,3] float64 on the GPU; returns optimised copies and an LMReport.
        `aux` (optional) carries host-side variables that decouple from the camera system (vector
        variables with only a prior factor): .error(), .try_lambda(lam) -> (lin, new), .accept()."""
        prm = params or LMParams()
        if prm.diagonalDamping:
            raise NotImplementedError("diagonalDamping=True is not implemented (gtsam default is False)")
        if not prm.useFixedLambdaFactor:
            raise NotImplementedError("useFixedLambdaFactor=False is not implemented (gtsam default is True)")
        poses = poses.to(torch.float64).contiguous().clone()
        points = points.to(torch.float64).contiguous().clone()
        rep = LMReport(setup_seconds=self.P.setup_seconds)
        torch.cuda.synchronize(self.P.device)
        t0 = time.perf_counter()
        lam = prm.lambdaInitial
        current = self.error(poses, points) + (aux.error() if aux else 0.0)
        rep.initial_error = current
        if current <= prm.errorTol or prm.maxIterations <= 0:
            rep.status, rep.final_error, rep.final_lambda = 0, current, lam
            return poses, points, rep
        while rep.iterations < prm.maxIterations:
            self.linearize(poses, points)                         # iterate(): linearise once
            new_error, stop_search, accepted = current, False, False
            lin0 = None
            while True:                                           # tryLambda
                self.schur(lam)
                self.band_solve()
                self.backsub()
                self.eval_step(poses, points)
                rec = self._trial.cpu()                           # the trial's ONE blocking read
                sc, status = rec[:4], int(rec[4:].view(torch.int32)[0])
                if status < 0:
                    if self._window_expired(status):              # the band is spoilt: redo this trial launch by launch
                        continue
                    raise RuntimeError("vus_ba_band_solve: the cooperative back-substitution timed out (status %d)" % status)
                rep.tries += 1
                a_lin, a_new = aux.try_lambda(lam) if aux else (0.0, 0.0)
                if lin0 is None:
                    lin0 = float(sc[0]) + (aux.error() if aux else 0.0)
                success = False
                if status == 0 and math.isfinite(float(sc[1])) and math.isfinite(float(sc[2])):
                    lin_change = lin0 - (float(sc[1]) + a_lin)
                    if lin_change >= 0.0:
                        new_err = float(sc[2]) + a_new
                        cost_change = current - new_err
                        if lin_change > 2.220446049250313e-16 * lin0:
                            success = cost_change / lin_change > prm.minModelFidelity
                        if abs(cost_change) < prm.relativeErrorTol * current:
                            stop_search = True
                        if success:
                            poses, self.new_poses = self.new_poses, poses
                            points, self.new_points = self.new_points, points
                            new_error = new_err
                            if aux:
                                aux.accept()
                if success:
                    lam = max(prm.lambdaLowerBound, lam / prm.lambdaFactor)
                    accepted = True
                    break
                if stop_search:
                    break
                lam *= prm.lambdaFactor
                if lam >= prm.lambdaUpperBound:
                    rep.status = 2
                    break
            rep.err_hist.append(new_error)
            rep.lambda_hist.append(lam)
            rep.outer += 1
            rep.iterations += int(accepted)
            if new_error <= prm.errorTol:
                converged = True
            else:
                abs_dec = current - new_error
                converged = (abs_dec / current <= prm.relativeErrorTol) or (abs_dec <= prm.absoluteErrorTol)
            current = new_error
            if rep.status == 2:
                break
            if converged:
                rep.status = 0
                break
            if not math.isfinite(current):
                break
        torch.cuda.synchronize(self.P.device)
        rep.seconds = time.perf_counter() - t0
        rep.final_error, rep.final_lambda = current, lam
        return poses, points, rep


# ---------------------------------------------------------------------------------------------
# graphs with inertial / velocity factors
class _CNav(ctypes.Structure):
    _fields_ = [("n_imu", c_int), ("imu_i", c_void_p), ("imu_j", c_void_p), ("imu_pim", c_void_p), ("imu_W", c_void_p),
                ("gravity", c_double * 3), ("n_dvl", c_int), ("dvl_pose", c_void_p), ("dvl_meas", c_void_p),
                ("dvl_w", c_void_p), ("n_vprior", c_int), ("vprior_idx", c_void_p), ("vprior_v", c_void_p),
                ("vprior_w", c_void_p)]


class NavFactors:
    """Device-resident vus_nav_factors.  imu = (i, j, pim [n,148], W [n,81]); dvl = (pose, meas [n,3], sigma [n]);
    vprior = (idx, v [n,3], sigmas [n,3]).  ImuFactors must join consecutive poses (j = i + 1)."""

    def __init__(self, gravity, imu=None, dvl=None, vprior=None, device="cuda:0"):
        dev = torch.device(device)

        def t(x, dt, shape):
            return torch.as_tensor(x).to(device=dev, dtype=dt).reshape(shape).contiguous()
        z = []
        self.imu_i = t(imu[0] if imu else z, torch.int32, (-1,))
        self.imu_j = t(imu[1] if imu else z, torch.int32, (-1,))
        self.imu_pim = t(imu[2] if imu else z, torch.float64, (-1, 148))
        self.imu_W = t(imu[3] if imu else z, torch.float64, (-1, 81))
        if self.imu_i.numel() and bool((self.imu_j - self.imu_i != 1).any()):
            raise NotImplementedError("ImuFactor between non-consecutive poses is not supported (batch.py:238 joins i-1 and i)")
        self.dvl_pose = t(dvl[0] if dvl else z, torch.int32, (-1,))
        self.dvl_meas = t(dvl[1] if dvl else z, torch.float64, (-1, 3))
        self.dvl_w = (1.0 / t(dvl[2], torch.float64, (-1,))).contiguous() if dvl else t(z, torch.float64, (-1,))
        self.vp_idx = t(vprior[0] if vprior else z, torch.int32, (-1,))
        self.vp_v = t(vprior[1] if vprior else z, torch.float64, (-1, 3))
        self.vp_w = (1.0 / t(vprior[2], torch.float64, (-1, 3))).contiguous() if vprior else t(z, torch.float64, (-1, 3))
        pp = lambda x: _lib.ptr(x) if x.numel() else None
        self.c = _CNav(self.imu_i.numel(), pp(self.imu_i), pp(self.imu_j), pp(self.imu_pim), pp(self.imu_W),
                       (c_double * 3)(*[float(g) for g in gravity]), self.dvl_pose.numel(), pp(self.dvl_pose),
                       pp(self.dvl_meas), pp(self.dvl_w), self.vp_idx.numel(), pp(self.vp_idx), pp(self.vp_v), pp(self.vp_w))
        self.n_factors = self.imu_i.numel() + self.dvl_pose.numel() + self.vp_idx.numel()

    def addr(self):
        return ctypes.addressof(self.c)


class NavBASolver(StereoBASolver):
    """LM over poses, velocities, one shared IMU bias and landmarks.  The problem must have been built with
    pose_stride=2 (velocity nodes interleaved); the bias is a 6-wide border eliminated after a
    7-right-hand-side band solve."""

    def __init__(self, problem: StereoBAProblem, nav: NavFactors):
        if problem.pose_stride != 2:
            raise ValueError("NavBASolver needs a StereoBAProblem built with pose_stride=2")
        super().__init__(problem)
        self.band_rhs = 7
        self._alloc_band_work()
        self.N = nav
        dev, nP, nN = problem.device, problem.n_poses, problem.n_nodes
        f64 = dict(dtype=torch.float64, device=dev)
        self.Snav = torch.empty((nN, 4, 36), **f64)
        self.Scb = torch.empty((nN, 36), **f64)
        self.Sbb = torch.empty((36,), **f64)
        self.gnav = torch.empty((nN, 6), **f64)
        self.gb = torch.empty((6,), **f64)
        self.rhs = torch.empty((7, nN * 6), **f64)
        self.db = torch.empty((6,), **f64)
        self.new_vels = torch.empty((nP, 3), **f64)
        self.new_bias = torch.empty((6,), **f64)
        self.nav_scal = torch.zeros((4,), **f64)
        self.nav_work = torch.empty((int(_lib.load().vus_nav_work_doubles(nav.addr())),), **f64)

    def nav_error(self, poses, vels, bias) -> float:
        p = _lib.ptr
        _lib.call("vus_nav_error", self.N.addr(), self.P.n_poses, p(poses), p(vels), p(bias), p(self.nav_scal),
                  p(self.nav_work), _lib.current_stream_ptr())
        return float(self.nav_scal[0].item())

    def nav_linearize(self, poses, vels, bias):
        p = _lib.ptr
        _lib.call("vus_nav_linearize", self.N.addr(), self.P.n_poses, p(poses), p(vels), p(bias), p(self.Snav),
                  p(self.Scb), p(self.Sbb), p(self.gnav), p(self.gb), p(self.nav_scal), p(self.nav_work),
                  _lib.current_stream_ptr())

    def nav_assemble(self, lam):
        p = _lib.ptr
        _lib.call("vus_nav_assemble", self.P.n_nodes, self.P.band, float(lam), p(self.Snav), p(self.Scb), p(self.gnav),
                  p(self.Sband), p(self.gs), p(self.rhs), _lib.current_stream_ptr())

    def nav_solve(self, lam):
        p = _lib.ptr
        st = _lib.current_stream_ptr()
        if self.use_split:
            _lib.call("vus_ba_band_solve_multi_split", p(self.Sband), self.P.n_nodes, self.P.band, p(self.rhs), 7,
                      p(self.status), p(self.band_work), st)
        else:
            _lib.call("vus_ba_band_solve_multi", p(self.Sband), self.P.n_nodes, self.P.band, p(self.rhs), 7, p(self.status), st)
        _lib.call("vus_nav_border_solve", self.P.n_nodes, p(self.rhs), p(self.Scb), p(self.Sbb), p(self.gb), float(lam),
                  p(self.dp), p(self.db), st)

    def nav_eval_step(self, poses, vels, bias):
        p = _lib.ptr
        _lib.call("vus_nav_eval_step", self.N.addr(), self.P.n_poses, p(poses), p(vels), p(bias), p(self.dp), p(self.db),
                  p(self.new_poses), p(self.new_vels), p(self.new_bias), p(self.nav_scal[1:]), p(self.nav_work),
                  _lib.current_stream_ptr())

    def optimize(self, poses, vels, bias, points, params: Optional[LMParams] = None):
        """Returns (poses, vels, bias, points, LMReport); inputs untouched."""
        prm = params or LMParams()
        if prm.diagonalDamping or not prm.useFixedLambdaFactor:
            raise NotImplementedError("only the gtsam defaults diagonalDamping=False, useFixedLambdaFactor=True")
        c = lambda x: x.to(torch.float64).contiguous().clone()
        poses, vels, bias, points = c(poses), c(vels), c(bias), c(points)
        rep = LMReport(setup_seconds=self.P.setup_seconds)
        torch.cuda.synchronize(self.P.device)
        t0 = time.perf_counter()
        lam = prm.lambdaInitial
        current = self.error(poses, points) + self.nav_error(poses, vels, bias)
        rep.initial_error = current
        if current <= prm.errorTol or prm.maxIterations <= 0:
            rep.status, rep.final_error, rep.final_lambda = 0, current, lam
            return poses, vels, bias, points, rep
        while rep.iterations < prm.maxIterations:
            self.linearize(poses, points)
            self.nav_linearize(poses, vels, bias)
            new_error, stop_search, accepted, lin0 = current, False, False, None
            while True:
                self.schur(lam)
                self.nav_assemble(lam)
                self.nav_solve(lam)
                self.backsub()
                self.eval_step(poses, points)
                self.nav_eval_step(poses, vels, bias)
                rec = self._trial.cpu()                           # stereo scalars + status, then the inertial scalars
                sc, status, nsc = rec[:4], int(rec[4:].view(torch.int32)[0]), self.nav_scal.cpu()
                if status < 0:
                    if self._window_expired(status):
                        continue
                    raise RuntimeError("vus_ba_band_solve: the cooperative back-substitution timed out (status %d)" % status)
                rep.tries += 1
                if lin0 is None:
                    lin0 = float(sc[0]) + float(nsc[0])
                success = False
                lin1, new1 = float(sc[1]) + float(nsc[1]), float(sc[2]) + float(nsc[2])
                if status == 0 and math.isfinite(lin1) and math.isfinite(new1):
                    lin_change = lin0 - lin1
                    if lin_change >= 0.0:
                        cost_change = current - new1
                        if lin_change > 2.220446049250313e-16 * lin0:
                            success = cost_change / lin_change > prm.minModelFidelity
                        if abs(cost_change) < prm.relativeErrorTol * current:
                            stop_search = True
                        if success:
                            poses, self.new_poses = self.new_poses, poses
                            points, self.new_points = self.new_points, points
                            vels, self.new_vels = self.new_vels, vels
                            bias, self.new_bias = self.new_bias, bias
                            new_error = new1
                if success:
                    lam = max(prm.lambdaLowerBound, lam / prm.lambdaFactor)
                    accepted = True
                    break
                if stop_search:
                    break
                lam *= prm.lambdaFactor
                if lam >= prm.lambdaUpperBound:
                    rep.status = 2
                    break
            rep.err_hist.append(new_error)
            rep.lambda_hist.append(lam)
            rep.outer += 1
            rep.iterations += int(accepted)
            if new_error <= prm.errorTol:
                converged = True
            else:
                abs_dec = current - new_error
                converged = (abs_dec / current <= prm.relativeErrorTol) or (abs_dec <= prm.absoluteErrorTol)
            current = new_error
            if rep.status == 2 or converged or not math.isfinite(current):
                if converged and rep.status != 2:
                    rep.status = 0
                break
        torch.cuda.synchronize(self.P.device)
        rep.seconds = time.perf_counter() - t0
        rep.final_error, rep.final_lambda = current, lam
        return poses, vels, bias, points, rep
