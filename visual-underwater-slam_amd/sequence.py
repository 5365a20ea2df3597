"""The reference's batch path end to end: stereo images -> front-end -> CameraMeasurement features -> get_landmarks
-> batch_update / batch_create -> LevenbergMarquardtOptimizer.optimize().

Host mirror of the parts of `AUV_ISAM` (/root/reference/batch.py) that lie on that path -- constants of batch.py:88-118,
`get_landmarks` (:144-176), `batch_update` (:253-266), `batch_create` (:270-305) and the optimiser call (:337) -- with
the ROS transport (subscribers, TF listener, time synchroniser; SURVEY.md out of scope) replaced by plain arguments.

Two ways from the front-end's output to the factor graph, producing the SAME graph:
  * `batch_update()` + `batch_create()`: the reference's own per-message / per-landmark Python loops, through the
    gtsam-shaped API (one GenericStereoFactor3D object per observation) -- what batch.py does, unchanged;
  * `batch_create_from_tracks()`: the whole stream at once -- `StereoOrbFrontend.stereo_factors` (vus_emit_stereo_factors)
    turns (ids, features, keyframe transforms) into factor arrays on the GPU, which enter the graph as ONE
    StereoFactorBlock and ONE insert_point3_block (first-sighting initialisation, keyframe 0 skipped, like the loop).

Two documented deviations from the reference's text, both switchable:
  * DVL factor: `gtsam.DvlVelocityFactor` (correct Jacobians) instead of the CustomFactor of batch.py:196-250, whose
    Jacobians are ill-formed.
  * `disparity_sign`: batch.py:156 computes `d = uR - uL`.  For a rig whose cam0 is the left camera that is negative, the
    triangulated point lands BEHIND the camera (z_cam = f * baseline / d < 0), every stereo factor then takes gtsam's
    cheirality branch (constant residual, zero Jacobians) and the landmarks never move.  `disparity_sign=-1` (default)
    is the reference verbatim; `disparity_sign=+1` evaluates the same formulas with the baseline's sign flipped
    (d / (-baseline) = (uL - uR) / baseline), which is the geometrically valid triangulation.  Both run the same kernel.
"""
from typing import List, Optional

import numpy as np
import torch

from . import gtsam
from .frontend import CameraMeasurement, ImageProcessorParams, StereoOrbFrontend, triangulate
from .gtsam.symbol_shorthand import B, L, V, X


def vector3(x, y, z):
    return np.array([x, y, z], dtype=float)


def depth_from_pressure(press_abs):
    """process_depth (batch.py:122-126): absolute pressure in hPa -> the depth that replaces the odometry's z
    (process_odom, batch.py:133-134).  Pinned by tests/golden/ref_batch_*.npz (the reference's own output)."""
    measured_pressure = press_abs * 100
    pressure_diff = measured_pressure - 98250.0
    return pressure_diff / (997 * 9.81)


class BatchSequence:
    """AUV_ISAM's batch accumulators and graph construction (batch.py:74-118, 144-176, 253-305)."""

    def __init__(self, disparity_sign: int = -1, device: str = "cuda:0"):
        assert disparity_sign in (-1, 1)
        self.device = torch.device(device)
        self.graph = gtsam.NonlinearFactorGraph()                                    # batch.py:80
        self.initial_estimate = gtsam.Values()                                       # :81
        self.timestep = 0
        # IMU (:87-92, 178-193)
        self.grav = 9.81
        self.g = np.array([0, 0, -self.grav])
        self.PARAMS = gtsam.PreintegrationParams.MakeSharedU(self.grav)
        I = np.eye(3)
        self.PARAMS.setAccelerometerCovariance(I * 8.999999999999999e-08)
        self.PARAMS.setGyroscopeCovariance(I * 1.2184696791468346e-07)
        self.PARAMS.setIntegrationCovariance(I * 1e-07)
        self.imu_preintegrated = gtsam.PreintegratedImuMeasurements(self.PARAMS)
        self.prev_bias = gtsam.imuBias.ConstantBias()
        # noise models (:95-98)
        self.pose_noise = gtsam.noiseModel.Diagonal.Sigmas(np.array([0.1, 0.1, 0.1, 0.3, 0.3, 0.3]))
        self.vel_noise = gtsam.noiseModel.Isotropic.Sigma(3, 0.1)
        self.dvl_noise = gtsam.noiseModel.Isotropic.Sigma(3, 0.1)
        # accumulators (:99-108)
        self.imu_data: List[np.ndarray] = []
        self.odom_accum, self.dvl_accum, self.imu_accum, self.landmark_accum = [], [], [], []
        self.zed_world_transform = None
        # camera (:110-118)
        self.baseline = 0.063
        self.intrinsic = [1827.0, 1827.5999755859375, 968.9000244140625, 561.4000244140625]
        self.f = (self.intrinsic[0] + self.intrinsic[1]) / 2.0
        self.cx, self.cy = self.intrinsic[2], self.intrinsic[3]
        self.K = gtsam.Cal3_S2Stereo(self.intrinsic[0], self.intrinsic[1], 0.0, self.cx, self.cy, self.baseline)
        self.resolution_x, self.resolution_y = 1920, 1080
        self.landmark_noise = gtsam.noiseModel.Isotropic.Sigma(3, 10)
        self.disparity_sign = disparity_sign
        self._tf_accum = []

    # -- camera ---------------------------------------------------------------------------------------------------
    def cam_array(self) -> torch.Tensor:
        """(fx, fy, cx, cy, baseline, resolution_x, resolution_y, 0) for vus_triangulate.  The reference's
        `d = uR - uL` (batch.py:156) is kept in the kernel; disparity_sign=+1 enters as a negated baseline."""
        b = self.baseline if self.disparity_sign < 0 else -self.baseline
        return torch.tensor([*self.intrinsic, b, self.resolution_x, self.resolution_y, 0.0], dtype=torch.float64,
                            device=self.device)

    def set_zed_world_transform(self, rot, trans):
        """What the TF callback stores (batch.py:45-48): (Rot3, translation)."""
        self.zed_world_transform = (rot, np.asarray(trans, dtype=float).reshape(3))

    def update_imu(self, acc, gyro):                                                 # batch.py:138-141
        self.imu_data.append(np.hstack((np.asarray(acc, float), np.asarray(gyro, float))))

    # -- batch.py:144-176 -----------------------------------------------------------------------------------------
    def get_landmarks(self, data: CameraMeasurement):
        landmarks = []
        if self.zed_world_transform is not None and len(data.features):              # :148
            feat = torch.tensor([[f.u0, f.v0, f.u1, f.v1] for f in data.features], dtype=torch.float64, device=self.device)
            Rt = torch.from_numpy(np.concatenate([self.zed_world_transform[0].matrix().reshape(-1),
                                                  self.zed_world_transform[1]])).to(self.device)
            out = triangulate(feat, self.cam_array(), Rt).cpu().numpy()              # :152-166 on the GPU
            for f, o in zip(data.features, out):
                landmarks.append({'id': f.id, 'pose': o[:3].copy(), 'uL': float(o[3]), 'uR': float(o[4]), 'v': float(o[5])})
        return landmarks

    # -- batch.py:253-266 -----------------------------------------------------------------------------------------
    def batch_update(self, odom_pose, dvl, landmarks: CameraMeasurement):
        """odom_pose: the Pose3 process_odom returns (:254); dvl: body-frame velocity 3-vector; landmarks: the
        CameraMeasurement message of this keyframe."""
        self.odom_accum.append(odom_pose)
        self.dvl_accum.append(np.asarray(dvl, dtype=float).reshape(3))
        self.imu_accum.append(self.imu_data)
        self.imu_data = []
        self._tf_accum.append(self.zed_world_transform)
        self.landmark_accum.append(self.get_landmarks(landmarks))

    # -- batch.py:270-305 -----------------------------------------------------------------------------------------
    def _camera_side(self):
        """Everything of batch_create except the landmark loop: priors, initial poses / velocities, IMU and DVL factors."""
        self.initial_estimate = gtsam.Values()
        self.graph = gtsam.NonlinearFactorGraph()
        self.initial_estimate.insert(B(0), self.prev_bias)                           # :274
        for i in range(len(self.odom_accum)):
            self.timestep = i
            pose = self.odom_accum[i]
            velocity = vector3(0, 0, 0)
            if i == 0:
                self.graph.add(gtsam.PriorFactorPose3(X(0), pose, self.pose_noise))  # :281
                self.graph.add(gtsam.PriorFactorVector(V(0), velocity, self.vel_noise))
                self.initial_estimate.insert(X(i), pose)
                self.initial_estimate.insert(V(i), velocity)
            else:
                self.initial_estimate.insert(X(i), pose)
                self.initial_estimate.insert(V(i), velocity)
                for imu in self.imu_accum[i]:
                    self.imu_preintegrated.integrateMeasurement(imu[:3], imu[3:], 0.005)   # :290
                self.graph.push_back(gtsam.ImuFactor(X(i - 1), V(i - 1), X(i), V(i), B(0), self.imu_preintegrated))
                self.graph.push_back(gtsam.DvlVelocityFactor(self.dvl_noise, V(i), X(i), self.dvl_accum[i]))
                self.imu_preintegrated.resetIntegration()                            # :293
                yield i

    def batch_create(self, with_landmark=True):
        """The reference's loop, object by object (batch.py:270-305)."""
        for i in self._camera_side():
            if with_landmark:
                for landmark in self.landmark_accum[i]:
                    if not self.initial_estimate.exists(L(landmark['id'])):          # :297
                        self.initial_estimate.insert(L(landmark['id']), landmark['pose'])
                    self.graph.push_back(gtsam.GenericStereoFactor3D(
                        gtsam.StereoPoint2(landmark['uL'], landmark['uR'], landmark['v']), self.landmark_noise,
                        X(i), L(landmark['id']), self.K))                            # :300-305

    def gate_factors(self, factors, Rt: torch.Tensor, gate_px: float):
        """EXTENSION (no counterpart in batch.py, whose input has been through the nodelet's RANSAC): drop every
        emitted factor whose residual AT THE INITIAL ESTIMATE -- keyframe transform Rt[f], first-sighting landmark --
        exceeds gate_px in any of (uL, uR, v) or whose landmark lies behind the camera (vus_stereo_initial_residuals).
        A brute-force Hamming mismatch is a residual of hundreds of pixels; without a robust kernel (gtsam's default,
        batch.py:118) one of them can push its landmark through a camera into the cheirality plateau.  Landmarks that
        lose every factor are dropped with them.  Returns the filtered dict (device tensors)."""
        n = factors["obs_frame"].numel()
        resid = torch.empty((n, 3), dtype=torch.float64, device=self.device)
        K = torch.from_numpy(self.K.vector6()).to(self.device)
        from . import _lib
        ptr = _lib.ptr
        _lib.call("vus_stereo_initial_residuals", ptr(Rt), ptr(K), ptr(factors["lm_point"]), ptr(factors["obs_frame"]),
                  ptr(factors["obs_id"]), ptr(factors["obs_meas"]), n, ptr(resid), _lib.current_stream_ptr())
        keep = resid.abs().amax(1) <= float(gate_px)
        out = dict(factors)
        for k in ("obs_frame", "obs_id", "obs_meas"):
            out[k] = factors[k][keep]
        still = torch.zeros_like(factors["lm_first"], dtype=torch.bool)
        still[out["obs_id"]] = True
        out["lm_first"] = torch.where(still, factors["lm_first"], torch.full_like(factors["lm_first"], -1))
        out["initial_residuals"], out["gate_keep"] = resid, keep
        return out

    def batch_create_from_tracks(self, factors):
        """EXTENSION: the same graph from the device arrays of StereoOrbFrontend.stereo_factors() -- one
        StereoFactorBlock in batch_create's factor order, one block of first-sighting landmark values."""
        for _ in self._camera_side():
            pass
        of, oi, om = factors["obs_frame"].cpu().numpy(), factors["obs_id"].cpu().numpy(), factors["obs_meas"].cpu().numpy()
        # keys by IN-PLACE adds: `X(0) + of.astype(...)` hands numpy a large temporary, and numpy decides whether it may
        # reuse it by walking the C stack (backtrace()): with the ROCm and torch libraries loaded the first such walk of a
        # process costs ~80 ms (measured: 78 of this function's 86 ms)
        if len(of):
            kx = of.astype(np.int64)
            kx += X(0)
            kl = oi.astype(np.int64)
            kl += L(0)
            self.graph.push_back(gtsam.StereoFactorBlock(om, self.landmark_noise, kx, kl, self.K))
        first = factors["lm_first"].cpu().numpy()
        seen = np.nonzero(first >= 0)[0]
        if len(seen):
            ks = seen.astype(np.int64)
            ks += L(0)
            self.initial_estimate.insert_point3_block(ks, factors["lm_point"].cpu().numpy()[seen])

    def optimize(self, params: Optional["gtsam.LevenbergMarquardtParams"] = None):
        """batch.py:337."""
        self.optimizer = gtsam.LevenbergMarquardtOptimizer(self.graph, self.initial_estimate,
                                                           params or gtsam.LevenbergMarquardtParams())
        return self.optimizer.optimize()


def keyframe_transforms(odom_poses) -> np.ndarray:
    """[F,12] zed_world_transform rows (row-major R then t) from a list of Pose3 / an [F,12] array."""
    if isinstance(odom_poses, np.ndarray):
        return np.ascontiguousarray(odom_poses, dtype=np.float64).reshape(-1, 12)
    return np.stack([p.flat12() for p in odom_poses])


# Front-end settings of the end-to-end sequence: the matcher is brute force (north star), so wrong temporal matches are
# kept out by a tight Hamming bound and the mutual-match filter rather than by the nodelet's RANSAC (stereo.launch:46).
SEQUENCE_PARAMS = dict(track_max_distance=30, cross_check=True)
GATE_PX = 60.0        # 6 sigma of the stereo noise model (batch.py:118: sigma = 10 px)


def run_sequence(frames: torch.Tensor, odom_poses: np.ndarray, imu, dvl, disparity_sign: int = 1,
                 params: Optional[ImageProcessorParams] = None, bulk: bool = True, frontend: Optional[StereoOrbFrontend] = None,
                 gate_px: float = GATE_PX):
    """Images to optimised trajectory: frames uint8 [F,2,H,W] on the GPU (one stereo pair per keyframe), odom_poses
    [F,12] (the odometry estimate of every keyframe = initial value of X(i) AND the camera transform get_landmarks
    uses), imu [F-1][n,>=6] samples between keyframes, dvl [F,3].  gate_px > 0 applies BatchSequence.gate_factors (bulk
    path only).  Returns (results Values, BatchSequence, stages dict)."""
    if params is None:
        params = ImageProcessorParams(**SEQUENCE_PARAMS)
    F, _, H, W = frames.shape
    fe = frontend or StereoOrbFrontend(H, W, max_frames=F, params=params)
    res = fe.process(frames)
    ids, feats, n_ids = fe.feature_tracks(res)
    seq = BatchSequence(disparity_sign=disparity_sign, device=str(frames.device))
    Rt = keyframe_transforms(odom_poses)
    stages = {"frontend": res, "ids": ids, "feats": feats, "n_ids": n_ids}
    if bulk:
        for i in range(F):
            seq.odom_accum.append(gtsam.Pose3.from_flat12(Rt[i]))
            seq.dvl_accum.append(np.asarray(dvl[i], float))
            seq.imu_accum.append([np.asarray(s, float)[:6] for s in imu[i - 1]] if i > 0 else [])
        Rt_d = torch.from_numpy(Rt).to(frames.device)
        factors = fe.stereo_factors(ids, feats, n_ids, Rt_d, seq.cam_array())
        stages["factors_ungated"] = factors
        if gate_px > 0:
            factors = seq.gate_factors(factors, Rt_d, gate_px)
        seq.batch_create_from_tracks(factors)
        stages["factors"] = factors
    else:
        msgs = fe.camera_measurements(res)
        for i in range(F):
            pose = gtsam.Pose3.from_flat12(Rt[i])
            seq.set_zed_world_transform(pose.rotation(), Rt[i, 9:])
            if i > 0:
                for s in imu[i - 1]:
                    seq.update_imu(s[:3], s[3:6])
            seq.batch_update(pose, dvl[i], msgs[i])
        seq.batch_create(with_landmark=True)
    results = seq.optimize()
    return results, seq, stages
