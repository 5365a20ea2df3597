"""Deterministic synthetic inputs for the hot path.

Everything is generated from counter-based 32-bit integer hashes evaluated with int64 array
arithmetic, so the SAME code runs under numpy (CPU tests, oracle inputs) and torch (on the GPU, for
the resident 1000-frame bench stream) and produces identical bytes.  No RNG stream of numpy or torch
is involved.

Stereo image stream (BASELINE.json configs[1]): a base canvas uint8[1024, 2048] of i.i.d. uniform
8x8-pixel blocks (corner-rich), seed 20261004; the left image of frame t is the 1280x720 canvas
window at offset (2t mod 700, t mod 280); the right image is the same scene shifted by an integer
disparity in 8..40 that is constant per 64-pixel column band; every pixel of every image gets its
own uniform noise in -4..4.
"""
import numpy as np

SEED = 20261004
CANVAS_H, CANVAS_W, BLOCK = 1024, 2048, 8
_M = 0xFFFFFFFF


def _mix32(x):
    """lowbias32 integer hash on int64 arrays holding values < 2**32 (numpy or torch)."""
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & _M
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & _M
    x = x ^ (x >> 16)
    return x


def _mix32_scalar(v):
    v &= _M
    v ^= v >> 16
    v = (v * 0x7FEB352D) & _M
    v ^= v >> 15
    v = (v * 0x846CA68B) & _M
    v ^= v >> 16
    return v


def _arange(xp, n, like=None):
    if xp is np:
        return np.arange(n, dtype=np.int64)
    return xp.arange(n, dtype=xp.int64, device=like)


def disparity_table(W, seed=SEED):
    """Integer disparity (8..40) of each 64-pixel column band of the right image."""
    n = (W + 63) // 64
    return [8 + _mix32_scalar(_mix32_scalar(seed) ^ (0xD15A0000 + b)) % 33 for b in range(n)]


def canvas(xp=np, device=None, seed=SEED):
    """uint8 canvas [1024, 2048] as an int64 array (values 0..255)."""
    by = _arange(xp, CANVAS_H, device) // BLOCK
    bx = _arange(xp, CANVAS_W, device) // BLOCK
    c = by[:, None] * (CANVAS_W // BLOCK) + bx[None, :]
    return _mix32(c ^ _mix32_scalar(seed)) & 255


def stereo_frames(t0, n_frames, H=720, W=1280, xp=np, device=None, seed=SEED, canvas_arr=None):
    """Frames t0 .. t0+n_frames-1 as uint8 [n_frames, 2, H, W] (index 1: 0 = left, 1 = right)."""
    assert H <= 720 + 24 and W <= 1280, "window must stay inside the canvas"
    cv = canvas(xp, device, seed) if canvas_arr is None else canvas_arr
    yy = _arange(xp, H, device)
    xx = _arange(xp, W, device)
    disp = disparity_table(W, seed)
    if xp is np:
        dcol = np.asarray([disp[x // 64] for x in range(W)], dtype=np.int64)
    else:
        dcol = xp.tensor([disp[x // 64] for x in range(W)], dtype=xp.int64, device=device)
    pix = yy[:, None] * W + xx[None, :]
    out = []
    for t in range(t0, t0 + n_frames):
        ox, oy = (2 * t) % 700, t % 280
        pair = []
        for cam in (0, 1):
            cols = ox + xx + (dcol if cam else 0)
            base = cv[oy:oy + H][:, cols]
            k = _mix32_scalar(_mix32_scalar((seed + t) & _M) ^ (0x9E3779B9 * (cam + 1)))
            noise = _mix32(pix ^ k) % 9 - 4
            img = base + noise
            img = img.clip(0, 255) if xp is np else img.clamp(0, 255)
            pair.append(img.astype(np.uint8) if xp is np else img.to(xp.uint8))
        out.append(np.stack(pair) if xp is np else xp.stack(pair))
    return np.stack(out) if xp is np else xp.stack(out)


# ---------------------------------------------------------------------------------------------
# Synthetic stereo bundle-adjustment sequences ("Synthetic BA input").
# Camera constants are the reference's: /root/reference/batch.py:110-118.
BASELINE_M = 0.063
INTRINSIC = (1827.0, 1827.5999755859375, 968.9000244140625, 561.4000244140625)  # fx, fy, cx, cy
RES_X, RES_Y = 1920, 1080
STEREO_SIGMA = 10.0
PRIOR_SIGMAS = (0.1, 0.1, 0.1, 0.3, 0.3, 0.3)


# BASELINE.json configs[2] ("2000 keyframes x 50k landmarks", SURVEY.md 8: 1000 observations per keyframe = 2 M factors) as
# arguments of ba_sequence(): landmarks nobody observes cannot enter a graph built like batch.py:295-305 and keyframes at
# the edge of the sweep see fewer than the cap, so 50 000 drawn / cap 1000 gave 48 299 landmarks and 1 926 616 factors
# (rounds 1-3).  52 000 drawn with a cap of 1032 gives 50 238 OBSERVED landmarks and 2 000 201 factors, band 224: the
# smallest such pair on this trajectory (tests/test_ba_c3_gpu.py asserts both counts).
CONFIGS2_BA = (2000, 52000, 1032)


def _hash_uniform(idx, key):
    """uniform (0,1) doubles from a counter array and a scalar key (numpy)."""
    h = _mix32(np.asarray(idx, dtype=np.int64) ^ _mix32_scalar(key))
    h2 = _mix32(h ^ 0x5BD1E995)
    return ((h * 4294967296.0 + h2) + 0.5) / 18446744073709551616.0


def _hash_normal(n, key):
    i = np.arange(n, dtype=np.int64)
    u1 = _hash_uniform(i, key)
    u2 = _hash_uniform(i, key ^ 0x68E31DA4)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def _rodrigues(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K


def ba_sequence(n_kf, n_lm, obs_per_kf, seed=SEED, line_len=None, kf_step=0.25, line_step=0.8,
                pose_sigma_t=0.05, pose_sigma_r=0.01, meas_sigma=1.0):
    """Lawn-mower sweep of a down-looking stereo rig at constant altitude over a slab of landmarks.

    Returns a dict of numpy arrays:
      poses_gt / poses_init [n_kf,12] (row-major R, then t; camera-to-world),
      points_gt / points_init [n_points,3],
      obs_pose, obs_point [n_obs] int32 and meas [n_obs,3] (uL,uR,v), sorted by (point, pose),
      K [6] (fx,fy,0,cx,cy,b), sigma, prior_sigmas.
    Every keyframe observes its (up to) obs_per_kf visible landmarks nearest the image centre;
    landmarks nobody observes are dropped and the rest renumbered.
    """
    fx, fy, cx, cy = INTRINSIC
    if line_len is None:
        line_len = max(5, int(round(np.sqrt(n_kf * line_step / kf_step))))
        line_len = min(line_len, 50)
    n_lines = (n_kf + line_len - 1) // line_len
    # trajectory: optical axis = world +Z, camera x = direction of travel
    poses = np.zeros((n_kf, 12))
    for i in range(n_kf):
        ln, k = divmod(i, line_len)
        fwd = ln % 2 == 0
        x = (k if fwd else line_len - 1 - k) * kf_step
        y = ln * line_step
        yaw = 0.0 if fwd else np.pi
        c, s = np.cos(yaw), np.sin(yaw)
        poses[i, :9] = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]]).reshape(-1)
        poses[i, 9:] = (x, y, 0.0)
    # landmarks: uniform over the swept footprint (+ margin), depth 2..6 m
    ext_x, ext_y = (line_len - 1) * kf_step, (n_lines - 1) * line_step
    j = np.arange(n_lm, dtype=np.int64)
    pts = np.stack([-2.0 + (ext_x + 4.0) * _hash_uniform(j, seed ^ 0x1111),
                    -1.2 + (ext_y + 2.4) * _hash_uniform(j, seed ^ 0x2222),
                    2.0 + 4.0 * _hash_uniform(j, seed ^ 0x3333)], 1)
    obs_p, obs_l, meas = [], [], []
    # Only landmarks inside the image footprint can be visible: |world dy| = |camera y| <= (RES_Y / fy) * depth.
    # Culling by a y-window first (landmarks sorted by y once) changes nothing in the output -- the candidates
    # keep their ascending index order -- and makes the 10 000-keyframe / 500 000-landmark sequence of
    # configs[4] a matter of seconds instead of minutes.
    y_order = np.argsort(pts[:, 1], kind="stable")
    y_sorted = pts[y_order, 1]
    y_reach = max(cy, RES_Y - cy) / fy * 6.0 + 1e-6
    for i in range(n_kf):
        R = poses[i, :9].reshape(3, 3)
        lo = np.searchsorted(y_sorted, poses[i, 10] - y_reach, "left")
        hi = np.searchsorted(y_sorted, poses[i, 10] + y_reach, "right")
        cand = np.sort(y_order[lo:hi])
        q = (pts[cand] - poses[i, 9:]) @ R    # rows: R^T (p - t)
        z = q[:, 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            uL = cx + fx * q[:, 0] / z
            uR = cx + fx * (q[:, 0] - BASELINE_M) / z
            v = cy + fy * q[:, 1] / z
        vis = (z > 0.5) & (uL >= 0) & (uL < RES_X) & (uR >= 0) & (uR < RES_X) & (v >= 0) & (v < RES_Y)
        idx = np.nonzero(vis)[0]
        if len(idx) > obs_per_kf:
            d2 = (uL[idx] - cx) ** 2 + (v[idx] - cy) ** 2
            idx = idx[np.argsort(d2, kind="stable")[:obs_per_kf]]
            idx.sort()
        obs_p.append(np.full(len(idx), i, np.int32))
        obs_l.append(cand[idx].astype(np.int32))
        meas.append(np.stack([uL[idx], uR[idx], v[idx]], 1))
    obs_p, obs_l, meas = np.concatenate(obs_p), np.concatenate(obs_l), np.concatenate(meas)
    used = np.unique(obs_l)
    remap = -np.ones(n_lm, np.int64)
    remap[used] = np.arange(len(used))
    obs_l = remap[obs_l].astype(np.int32)
    pts = pts[used]
    order = np.lexsort((obs_p, obs_l))
    obs_p, obs_l, meas = obs_p[order], obs_l[order], meas[order]
    n_obs = len(obs_p)
    meas = meas + meas_sigma * _hash_normal(3 * n_obs, seed ^ 0x4444).reshape(n_obs, 3)
    # initial estimate: perturbed poses (X0 exact = its prior, like the odometry prior of batch.py:281-283)
    poses_init = poses.copy()
    nt = pose_sigma_t * _hash_normal(3 * n_kf, seed ^ 0x5555).reshape(n_kf, 3)
    nr = pose_sigma_r * _hash_normal(3 * n_kf, seed ^ 0x6666).reshape(n_kf, 3)
    for i in range(1, n_kf):
        R = poses[i, :9].reshape(3, 3) @ _rodrigues(nr[i])
        poses_init[i, :9] = R.reshape(-1)
        poses_init[i, 9:] = poses[i, 9:] + nt[i]
    # landmark initial estimate: stereo triangulation of the first observation from the perturbed pose
    first = np.concatenate([[0], np.nonzero(np.diff(obs_l))[0] + 1])
    m0, p0 = meas[first], obs_p[first]
    disp = np.maximum(m0[:, 0] - m0[:, 1], 0.5)
    zc = fx * BASELINE_M / disp
    cam = np.stack([(m0[:, 0] - cx) * zc / fx, (m0[:, 2] - cy) * zc / fy, zc], 1)
    Rm = poses_init[p0, :9].reshape(-1, 3, 3)
    pts_init = np.einsum("nij,nj->ni", Rm, cam) + poses_init[p0, 9:]
    return {
        "poses_gt": poses, "poses_init": poses_init, "points_gt": pts, "points_init": pts_init,
        "obs_pose": obs_p.astype(np.int32), "obs_point": obs_l.astype(np.int32), "meas": meas,
        "K": np.array([fx, fy, 0.0, cx, cy, BASELINE_M]), "sigma": STEREO_SIGMA,
        "prior_sigmas": np.array(PRIOR_SIGMAS), "line_len": line_len,
    }


# ---------------------------------------------------------------------------------------------
# Stereo + IMU + DVL sequence (BASELINE.json configs[0]; IMU constants of /root/reference/batch.py:88,183-185,290)
IMU_DT = 0.005
GRAVITY = 9.81
IMU_ACC_COV, IMU_GYRO_COV, IMU_INT_COV = 8.999999999999999e-08, 1.2184696791468346e-07, 1e-07


def nav_sequence(n_kf, n_lm, obs_per_kf, seed=SEED, kf_period=0.2, pose_sigma_t=0.05, pose_sigma_r=0.01,
                 meas_sigma=1.0, dvl_sigma=0.0, rest_start=False):
    """Down-looking stereo rig on a smooth meandering track with a 200 Hz IMU (dt = 0.005, batch.py:290)
    and a DVL (body-frame velocity).  The keyframe states are produced by integrating the sampled IMU
    signals with the same discrete model the preintegration uses, so the inertial factors are exactly
    consistent with the ground truth; the images see landmarks 2-6 m below the vehicle.

    Returns the ba_sequence() dict plus: vels_gt [n_kf,3], imu [n_kf-1, n_per, 7] (ax,ay,az,wx,wy,wz,dt),
    dvl [n_kf,3], gravity [3]."""
    from .gtsam.imu import so3_expmap
    n_per = int(round(kf_period / IMU_DT))
    g = np.array([0.0, 0.0, -GRAVITY])
    # desired smooth motion (analytic), camera/body z axis pointing down
    R0 = np.diag([1.0, -1.0, -1.0])

    def desired(t):
        yaw = 0.35 * np.sin(0.15 * t)
        pitch, roll = 0.04 * np.sin(0.5 * t), 0.03 * np.sin(0.37 * t + 1.0)
        cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
        Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
        Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
        Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
        if rest_start:      # the vehicle starts from rest, as batch.py:282's zero-velocity prior on V(0) assumes
            p = np.array([1.25 * (t - 1.5 * (1.0 - np.exp(-t / 1.5))), 1.5 * (1.0 - np.cos(0.2 * t)),
                          0.15 * (1.0 - np.cos(0.3 * t))])
        else:
            p = np.array([1.25 * t, 1.5 * np.sin(0.2 * t), 0.15 * np.sin(0.3 * t)])
        return Rz @ Ry @ Rx @ R0, p

    def derivs(t, h=1e-4):
        Rm, pm = desired(t - h); Rp, pp = desired(t + h); Rc, pc = desired(t)
        vel = (pp - pm) / (2 * h)
        acc = (pp - 2 * pc + pm) / (h * h)
        dR = Rc.T @ (Rp - Rm) / (2 * h)
        om = np.array([dR[2, 1] - dR[1, 2], dR[0, 2] - dR[2, 0], dR[1, 0] - dR[0, 1]]) * 0.5
        return Rc, pc, vel, om, Rc.T @ (acc - g)

    Rk, pk, vk, _, _ = derivs(0.0)
    poses = np.zeros((n_kf, 12)); vels = np.zeros((n_kf, 3))
    imu = np.zeros((max(n_kf - 1, 0), n_per, 7))
    poses[0, :9], poses[0, 9:], vels[0] = Rk.reshape(-1), pk, vk
    for i in range(1, n_kf):
        for s in range(n_per):
            t = ((i - 1) * n_per + s + 0.5) * IMU_DT          # mid-interval sample of the smooth signals
            _, _, _, om, ab = derivs(t)
            imu[i - 1, s] = [*ab, *om, IMU_DT]
            pk = pk + vk * IMU_DT + 0.5 * g * IMU_DT ** 2 + 0.5 * (Rk @ ab) * IMU_DT ** 2
            vk = vk + g * IMU_DT + (Rk @ ab) * IMU_DT
            Rk = Rk @ so3_expmap(om * IMU_DT)
        poses[i, :9], poses[i, 9:], vels[i] = Rk.reshape(-1), pk, vk
    # landmarks below the track
    fx, fy, cx, cy = INTRINSIC
    j = np.arange(n_lm, dtype=np.int64)
    x_max = poses[-1, 9] + 3.0
    pts = np.stack([-3.0 + (x_max + 3.0) * _hash_uniform(j, seed ^ 0x1111),
                    -4.5 + 9.0 * _hash_uniform(j, seed ^ 0x2222),
                    -(2.0 + 4.0 * _hash_uniform(j, seed ^ 0x3333))], 1)
    obs_p, obs_l, meas = [], [], []
    # footprint cull along the track (see ba_sequence): |world dx| <= 960/fx * 6.2 * cos + 540/fy * 6.2 * sin < 4.5 m
    x_order = np.argsort(pts[:, 0], kind="stable")
    x_sorted = pts[x_order, 0]
    for i in range(n_kf):
        R = poses[i, :9].reshape(3, 3)
        lo = np.searchsorted(x_sorted, poses[i, 9] - 4.5, "left")
        hi = np.searchsorted(x_sorted, poses[i, 9] + 4.5, "right")
        cand = np.sort(x_order[lo:hi])
        q = (pts[cand] - poses[i, 9:]) @ R
        z = q[:, 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            uL = cx + fx * q[:, 0] / z
            uR = cx + fx * (q[:, 0] - BASELINE_M) / z
            v = cy + fy * q[:, 1] / z
        vis = (z > 0.5) & (uL >= 0) & (uL < RES_X) & (uR >= 0) & (uR < RES_X) & (v >= 0) & (v < RES_Y)
        idx = np.nonzero(vis)[0]
        if len(idx) > obs_per_kf:
            d2 = (uL[idx] - cx) ** 2 + (v[idx] - cy) ** 2
            idx = np.sort(idx[np.argsort(d2, kind="stable")[:obs_per_kf]])
        obs_p.append(np.full(len(idx), i, np.int32)); obs_l.append(cand[idx].astype(np.int32))
        meas.append(np.stack([uL[idx], uR[idx], v[idx]], 1))
    obs_p, obs_l, meas = np.concatenate(obs_p), np.concatenate(obs_l), np.concatenate(meas)
    used = np.unique(obs_l)
    remap = -np.ones(n_lm, np.int64); remap[used] = np.arange(len(used))
    obs_l = remap[obs_l].astype(np.int32); pts = pts[used]
    order = np.lexsort((obs_p, obs_l))
    obs_p, obs_l, meas = obs_p[order], obs_l[order], meas[order]
    n_obs = len(obs_p)
    meas = meas + meas_sigma * _hash_normal(3 * n_obs, seed ^ 0x4444).reshape(n_obs, 3)
    poses_init = poses.copy()
    nt = pose_sigma_t * _hash_normal(3 * n_kf, seed ^ 0x5555).reshape(n_kf, 3)
    nr = pose_sigma_r * _hash_normal(3 * n_kf, seed ^ 0x6666).reshape(n_kf, 3)
    for i in range(1, n_kf):
        poses_init[i, :9] = (poses[i, :9].reshape(3, 3) @ _rodrigues(nr[i])).reshape(-1)
        poses_init[i, 9:] = poses[i, 9:] + nt[i]
    first = np.concatenate([[0], np.nonzero(np.diff(obs_l))[0] + 1])
    m0, p0 = meas[first], obs_p[first]
    disp = np.maximum(m0[:, 0] - m0[:, 1], 0.5)
    zc = fx * BASELINE_M / disp
    cam = np.stack([(m0[:, 0] - cx) * zc / fx, (m0[:, 2] - cy) * zc / fy, zc], 1)
    pts_init = np.einsum("nij,nj->ni", poses_init[p0, :9].reshape(-1, 3, 3), cam) + poses_init[p0, 9:]
    dvl = np.einsum("nji,nj->ni", poses[:, :9].reshape(-1, 3, 3), vels)       # R^T v
    if dvl_sigma > 0:
        dvl = dvl + dvl_sigma * _hash_normal(3 * n_kf, seed ^ 0x7777).reshape(n_kf, 3)
    return {
        "poses_gt": poses, "poses_init": poses_init, "points_gt": pts, "points_init": pts_init,
        "vels_gt": vels, "imu": imu, "dvl": dvl, "gravity": g,
        "obs_pose": obs_p.astype(np.int32), "obs_point": obs_l.astype(np.int32), "meas": meas,
        "K": np.array([fx, fy, 0.0, cx, cy, BASELINE_M]), "sigma": STEREO_SIGMA, "prior_sigmas": np.array(PRIOR_SIGMAS),
    }


# ---------------------------------------------------------------------------------------------
# A RENDERED scene: the end-to-end sequence of the north star (images -> front-end -> feature tracks ->
# get_landmarks -> batch_create -> LM; /root/reference/batch.py:144-176, 253-266, 270-305, 337).
# A textured sea floor (the plane z = SCENE_PLANE_Z of the world frame, random 26-mm blocks: corner-rich like the
# canvas above) seen by the down-looking rectified stereo rig of nav_sequence(): every pixel is the intersection of its
# viewing ray with the plane, so disparity, parallax and perspective are those of the real geometry.  The calibration
# is the reference's (batch.py:111, given for the 1920 x 1080 frame that get_landmarks' mapping u = (u0 + 1) / 2 * 1920
# assumes, batch.py:116-117,152-154): pixel x of a W-pixel-wide image is the ray through u = x * 1920 / W.
# Plain float64 multiply / add / divide / floor and the integer hash only: the same code under numpy and torch.
SCENE_PLANE_Z = -4.0
SCENE_TEXEL = 0.026


def render_plane_view(T12, cam_offset_x, t, cam, H=720, W=1280, xp=np, device=None, seed=SEED):
    """uint8 [H, W] view of the textured plane from the camera with pose T12 (row-major R then t, camera-to-world)
    displaced by cam_offset_x along its own x axis (0 = left / cam0, the baseline = right / cam1).  `t`, `cam` key the
    per-pixel sensor noise (uniform -4..4) like stereo_frames()."""
    fx, fy, cx, cy = INTRINSIC
    R = [float(v) for v in T12[:9]]
    o = [float(T12[9 + r]) + R[3 * r] * float(cam_offset_x) for r in range(3)]
    if xp is np:
        xs = np.arange(W, dtype=np.float64)[None, :]
        ys = np.arange(H, dtype=np.float64)[:, None]
    else:
        xs = xp.arange(W, dtype=xp.float64, device=device)[None, :]
        ys = xp.arange(H, dtype=xp.float64, device=device)[:, None]
    xn = (xs * (RES_X / W) - cx) / fx
    yn = (ys * (RES_Y / H) - cy) / fy
    dz = (xn * R[6] + yn * R[7]) + R[8]
    lam = (SCENE_PLANE_Z - o[2]) / dz
    px = ((xn * R[0] + yn * R[1]) + R[2]) * lam + o[0]
    py = ((xn * R[3] + yn * R[4]) + R[5]) * lam + o[1]
    if xp is np:
        bx = np.floor(px / SCENE_TEXEL).astype(np.int64) + 32768
        by = np.floor(py / SCENE_TEXEL).astype(np.int64) + 32768
    else:
        bx = xp.floor(px / SCENE_TEXEL).to(xp.int64) + 32768
        by = xp.floor(py / SCENE_TEXEL).to(xp.int64) + 32768
    base = _mix32((((bx & 0xFFFF) << 16) | (by & 0xFFFF)) ^ _mix32_scalar(seed ^ 0x5CE7E)) & 255
    pix = _arange(xp, H, device)[:, None] * W + _arange(xp, W, device)[None, :]
    k = _mix32_scalar(_mix32_scalar((seed + t) & _M) ^ (0x9E3779B9 * (cam + 1)))
    img = base + (_mix32(pix ^ k) % 9 - 4)
    img = img.clip(0, 255) if xp is np else img.clamp(0, 255)
    return img.astype(np.uint8) if xp is np else img.to(xp.uint8)


def scene_frames(poses, H=720, W=1280, xp=np, device=None, seed=SEED):
    """Rectified stereo pairs uint8 [n, 2, H, W] of the plane from the camera poses [n, 12]."""
    out = []
    for t in range(len(poses)):
        pair = [render_plane_view(poses[t], BASELINE_M * c, t, c, H, W, xp, device, seed) for c in (0, 1)]
        out.append(np.stack(pair) if xp is np else xp.stack(pair))
    return np.stack(out) if xp is np else xp.stack(out)


def scene_sequence(n_kf, H=720, W=1280, seed=SEED, render=True, **nav_kw):
    """nav_sequence()'s trajectory, IMU and DVL data plus the rendered stereo pair of every keyframe ("frames",
    uint8 [n_kf, 2, H, W]).  The sequence's synthetic landmark observations are NOT used by the end-to-end chain: its
    landmarks come out of the front-end."""
    nav_kw.setdefault("rest_start", True)
    s = nav_sequence(n_kf, 300, 12, seed=seed, **nav_kw)
    s["image_size"] = (H, W)
    if render:
        s["frames"] = scene_frames(s["poses_gt"], H, W, seed=seed)
    return s
