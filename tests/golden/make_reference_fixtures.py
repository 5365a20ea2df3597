"""Golden vectors produced BY THE REFERENCE'S OWN CODE (tests/golden/ref_batch_*.npz).  BUILD CONTAINER ONLY.

    python tests/golden/make_reference_fixtures.py          # needs /root/reference; no GPU; ~5 s

What runs: /root/reference/batch.py, loaded by path and unmodified, driven through its own entry points -- the
callbacks `imu_callback`, `pressure_callback`, `ts_callback` (batch.py:32-55) and `AUV_ISAM.process_depth` (:122-126),
`process_odom` (:128-136), `update_imu` (:138-141), `get_landmarks` (:144-176), `batch_update` (:253-266),
`batch_create(True)` (:270-305), `constr3DPoints` (:57-68) and the MSE statements of its `__main__` block (:351-353, 362-366,
selected from the file's syntax tree at run time and executed; no line of the reference is stored anywhere).

What is stubbed, and why that does not touch the arithmetic being pinned:
  * TRANSPORT / PLOT modules that are not installed here and hold no arithmetic of the path: rospy, rosnode, tf2_ros,
    message_filters, std_msgs, sensor_msgs, nav_msgs, geometry_msgs, waterlinked_a50_ros_driver, gtsam_vio (message
    classes only).  They are empty namespaces below; messages are plain attribute bags.
  * `gtsam` is not installed either: the name is mapped to THIS repository's gtsam-shaped module
    (visual_underwater_slam_amd.gtsam, host-side containers and geometry classes only -- no kernel, no GPU).  So what
    these fixtures pin is everything batch.py computes ITSELF, in numpy: the NDC -> pixel mapping, disparity, back-projection
    and rigid transform of get_landmarks, the pressure -> depth formula, the order / keys / measurements batch_create
    pushes, which landmark value is the first sighting, constr3DPoints' rows and the MSE.  Quantities that pass through
    the shim's classes (Rot3.Quaternion -> matrix, Pose3 accessors, the symbol keys) are the shim's own output and are
    marked "via shim" in the README below; GTSAM's arithmetic (factors, LM) stays unpinned.

What is written: DATA ONLY -- inputs and the values the reference produced.  The reference never travels to the GPU box.
The input stream is the oracle front-end's output (oracle/chain.py) on synth.scene_sequence: real feature tracks with
persistent ids, plus two hand-made edge cases (a keyframe arriving before any TF, an empty feature list).
"""
import ast
import importlib.util
import json
import os
import sys
import types
from functools import partial

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference/batch.py"
sys.path.insert(0, ROOT)


class Bag(types.SimpleNamespace):
    """A ROS message as batch.py reads it: nested attributes."""


def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _TfError(Exception):
        pass

    mod("rospy", Time=lambda *a: 0, logerr=lambda *a, **k: None, init_node=lambda *a, **k: None)
    mod("rosnode")
    mod("tf2_ros", LookupException=type("LookupException", (_TfError,), {}),
        ConnectivityException=type("ConnectivityException", (_TfError,), {}),
        ExtrapolationException=type("ExtrapolationException", (_TfError,), {}))
    mod("message_filters")
    for pkg, names in (("std_msgs", ["String"]), ("sensor_msgs", ["Imu"]), ("nav_msgs", ["Odometry"]),
                       ("geometry_msgs", ["PoseWithCovarianceStamped", "TwistStamped"]),
                       ("waterlinked_a50_ros_driver", ["DVL"]), ("gtsam_vio", ["CameraMeasurement"])):
        mod(pkg)
        mod(pkg + ".msg", **{n: Bag for n in names})
    import visual_underwater_slam_amd.gtsam as shim            # host containers only; nothing here loads the HIP library
    sys.modules["gtsam"] = shim
    sys.modules["gtsam.symbol_shorthand"] = shim.symbol_shorthand
    sys.modules["gtsam.utils"] = shim.utils
    sys.modules["gtsam.utils.plot"] = shim.utils.plot
    return shim


def load_reference():
    shim = _install_stubs()
    spec = importlib.util.spec_from_file_location("reference_batch", REFERENCE)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)                                 # defines functions + AUV_ISAM; the __main__ block does not run
    return ref, shim


def main_block_statements(targets):
    """The statements of the reference's `if __name__ == '__main__':` block that assign one of `targets`, compiled from
    the file's own syntax tree (batch.py:351-353, 362-366: x, y, z, odom, squared_diff, mse)."""
    tree = ast.parse(open(REFERENCE).read(), REFERENCE)
    main = [n for n in tree.body if isinstance(n, ast.If) and "__main__" in ast.dump(n.test)][0]
    keep = []
    for st in main.body:
        names = []
        if isinstance(st, ast.Assign):
            names = [t.id for t in st.targets if isinstance(t, ast.Name)]
        elif isinstance(st, ast.AugAssign) and isinstance(st.target, ast.Name):
            names = [st.target.id]
        if any(n in targets for n in names):
            keep.append(st)
    return compile(ast.Module(body=keep, type_ignores=[]), REFERENCE, "exec")


def quat_wxyz(R):
    """Unit quaternion of a rotation matrix (Shepperd), for the fake TF / odometry messages."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = [0.0] * 4
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    return np.array(q) / np.linalg.norm(q)


def feature_stream(F, H, W, KP):
    """The oracle front-end on the rendered scene: (ids [F,K], feats [F,K,4] = u0 v0 u1 v1, n_ids, scene)."""
    from oracle import chain
    from visual_underwater_slam_amd import sequence, synth
    s = synth.scene_sequence(F, H, W)
    fe = chain.frontend(s["frames"], KP, **sequence.SEQUENCE_PARAMS)
    return fe["ids"], fe["feats"], fe["n_ids"], s


def drive_reference(ref, shim, ids, feats, scene, tf_missing_first=True):
    """Feed the stream through the reference's callbacks; returns the AUV_ISAM object and what went in."""
    F, K = ids.shape
    slam = ref.AUV_ISAM()
    ref.slam = slam                                              # the callbacks use module globals (batch.py:32-55)
    tf_now = {}

    class FakeTfBuffer:
        def lookup_transform(self, target, source, time):
            if "q" not in tf_now:
                raise sys.modules["tf2_ros"].LookupException("no transform yet")
            q, t = tf_now["q"], tf_now["t"]
            return Bag(transform=Bag(translation=Bag(x=t[0], y=t[1], z=t[2]), rotation=Bag(w=q[0], x=q[1], y=q[2], z=q[3])))

    ref.tf_buffer = FakeTfBuffer()
    poses = scene["poses_init"]
    inputs = dict(press_abs=[], odom_xyz=[], odom_quat=[], tf_quat=[], tf_trans=[], dvl=[], imu_count=[], has_tf=[])
    for i in range(F):
        R, t = poses[i, :9].reshape(3, 3), poses[i, 9:]
        if i > 0:                                                # IMU samples of the interval, one callback each
            for smp in scene["imu"][i - 1]:
                ref.imu_callback(Bag(linear_acceleration=Bag(x=smp[0], y=smp[1], z=smp[2]),
                                     angular_velocity=Bag(x=smp[3], y=smp[4], z=smp[5]), header=Bag(stamp=i)))
        inputs["imu_count"].append(0 if i == 0 else len(scene["imu"][i - 1]))
        # pressure whose depth is (close to) the pose's z: press_abs in hPa (batch.py:122-126)
        press = (float(t[2]) * (997 * 9.81) + 98250.0) / 100.0
        ref.pressure_callback(Bag(data=json.dumps({"press_abs": press})))
        inputs["press_abs"].append(press)
        q = quat_wxyz(R)
        if not (tf_missing_first and i < 2):                     # keyframes 0 and 1 arrive before the first TF: no landmarks (:148)
            tf_now["q"], tf_now["t"] = q, t
        inputs["has_tf"].append("q" in tf_now)
        inputs["tf_quat"].append(q); inputs["tf_trans"].append(t)
        odom = Bag(pose=Bag(pose=Bag(position=Bag(x=t[0], y=t[1], z=t[2] + 0.25),      # z is replaced by depth (:133-134)
                                      orientation=Bag(w=q[0], x=q[1], y=q[2], z=q[3]))))
        inputs["odom_xyz"].append([t[0], t[1], t[2] + 0.25]); inputs["odom_quat"].append(q)
        d = scene["dvl"][i]
        dvl = Bag(twist=Bag(linear=Bag(x=d[0], y=d[1], z=d[2])))
        inputs["dvl"].append(d)
        feats_i = [Bag(id=int(ids[i, k]), u0=float(feats[i, k, 0]), v0=float(feats[i, k, 1]), u1=float(feats[i, k, 2]),
                       v1=float(feats[i, k, 3])) for k in np.nonzero(ids[i] >= 0)[0]]
        ref.ts_callback(odom, dvl, Bag(features=feats_i))
    return slam, {k: np.array(v) for k, v in inputs.items()}


def capture(ref, shim, slam, ids, feats, inputs):
    from visual_underwater_slam_amd.gtsam.symbol_shorthand import symbolChr, symbolIndex
    F, K = ids.shape
    out = dict(ids=ids, feats=feats, **{"in_" + k: v for k, v in inputs.items()})
    # ---- get_landmarks (batch.py:144-176), one row per feature in message order
    lf, lid, lpose, lm = [], [], [], []
    for i, lms in enumerate(slam.landmark_accum):
        for d in lms:
            lf.append(i); lid.append(d["id"]); lpose.append(d["pose"]); lm.append([d["uL"], d["uR"], d["v"]])
    out.update(lm_frame=np.array(lf, np.int32), lm_id=np.array(lid, np.int64), lm_pose=np.array(lpose).reshape(-1, 3),
               lm_meas=np.array(lm).reshape(-1, 3))
    out["tf_matrix"] = np.array([np.concatenate([shim.Rot3.Quaternion(*q).matrix().reshape(-1), t])      # via shim
                                 for q, t in zip(inputs["tf_quat"], inputs["tf_trans"])])
    out["depth"] = np.array([(p * 100 - 98250.0) / (997 * 9.81) for p in inputs["press_abs"]])          # restated :122-126 ...
    out["odom_adjust"] = np.array([p.flat12() for p in slam.odom_accum])                                 # ... and what process_odom stored
    out["odom_compare"] = np.array([p.flat12() for p in slam.odom_compare])
    assert np.array_equal(out["odom_adjust"][:, 11], out["depth"]), "process_depth / process_odom: depth is the pose's z"
    # ---- batch_create(True) (batch.py:270-305): the graph in push order
    slam.batch_create(with_landmark=True)
    g = slam.graph
    ftype, fkeys = [], []
    st_meas, st_pk, st_lk, dvl_meas, dvl_keys, imu_keys, imu_dt, imu_dR, imu_dp, imu_dv = [], [], [], [], [], [], [], [], [], []
    code = {"PriorFactorPose3": 0, "PriorFactorVector": 1, "ImuFactor": 2, "CustomFactor": 3, "GenericStereoFactor3D": 4}
    for n in range(g.size()):
        f = g.at(n)
        name = type(f).__name__
        ftype.append(code[name])
        ks = list(f.keys())
        fkeys.append(ks + [-1] * (5 - len(ks)))
        if name == "GenericStereoFactor3D":
            st_meas.append(f.measured().vector()); st_pk.append(ks[0]); st_lk.append(ks[1])
        elif name == "CustomFactor":
            fn = f._fn
            assert isinstance(fn, partial) and len(fn.args) == 1, "batch.py:245-249: partial(self.velocity_error, measurement)"
            dvl_meas.append(np.asarray(fn.args[0]).reshape(3)); dvl_keys.append(ks)
        elif name == "ImuFactor":
            imu_keys.append(ks)
    out.update(factor_type=np.array(ftype, np.int8), factor_keys=np.array(fkeys, np.int64),
               stereo_meas=np.array(st_meas).reshape(-1, 3), stereo_pose_key=np.array(st_pk, np.int64),
               stereo_lm_key=np.array(st_lk, np.int64), dvl_meas=np.array(dvl_meas).reshape(-1, 3),
               dvl_keys=np.array(dvl_keys, np.int64).reshape(-1, 2), imu_keys=np.array(imu_keys, np.int64).reshape(-1, 5))
    # ---- the initial estimate
    v = slam.initial_estimate
    keys = np.array(v.keys(), np.int64)
    out["value_keys"] = keys
    chr_ = np.array([symbolChr(int(k)) for k in keys])
    lk = keys[chr_ == "l"]
    out["value_lm_key"] = lk
    out["value_lm_point"] = np.array([v.atPoint3(int(k)) for k in lk]).reshape(-1, 3)
    xk = keys[chr_ == "x"]
    out["value_pose_key"] = xk
    out["value_pose"] = np.array([v.atPose3(int(k)).flat12() for k in xk])
    out["value_vel"] = np.array([v.atVector(int(k)) for k in keys[chr_ == "v"]])
    # ---- reporting: constr3DPoints on a Values holding known poses, then the MSE statements (batch.py:57-68, 351-366)
    res = shim.Values()
    from visual_underwater_slam_amd.gtsam.symbol_shorthand import X
    rng = np.random.default_rng(7)
    rep_poses = out["odom_adjust"].copy()
    rep_poses[:, 9:] += 0.05 * rng.standard_normal((F, 3))
    for i in range(F):
        res.insert(X(i), shim.Pose3.from_flat12(rep_poses[i]))
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):              # constr3DPoints prints values.exists(i) (:66)
        points = ref.constr3DPoints(res)
    ns = dict(np=np, slam=slam, points=points)
    exec(main_block_statements({"x", "y", "z", "odom", "squared_diff", "mse"}), ns)
    out.update(report_poses=rep_poses, report_points=points[1:], report_mse=np.array(ns["mse"]),
               report_odom_shifted=ns["odom"])
    # ---- does optimize() recognise the reference's own DVL CustomFactor objects?
    from visual_underwater_slam_amd.gtsam.optimizer import lower_reference_dvl_factor
    lowered = [lower_reference_dvl_factor(g.at(n)) for n in range(g.size()) if ftype[n] == 3]
    assert all(l is not None for l in lowered) and len(lowered) == F - 1
    out["dvl_lowered_meas"] = np.array([l.measured for l in lowered]).reshape(-1, 3)
    return out


README = """ref_batch_*.npz -- produced by tests/golden/make_reference_fixtures.py from /root/reference/batch.py (unmodified)
in: ids, feats (CameraMeasurement stream: id, u0 v0 u1 v1 per keyframe slot; id -1 = empty slot), in_* (message fields)
reference-computed (numpy inside batch.py): lm_frame, lm_id, lm_pose, lm_meas (get_landmarks), depth/odom_adjust z
   (process_depth), factor_type/factor_keys order, stereo_* and dvl_* columns, value_lm_* (first sighting), report_*
via shim (this repo's gtsam-shaped classes, called by the reference): tf_matrix, odom_* rotations, symbol keys
"""


def main():
    ref, shim = load_reference()
    cases = {
        # name: (F, H, W, KP, tf_missing_first)
        "scene6": (6, 360, 640, 300, False),
        "scene5_late_tf": (5, 240, 320, 120, True),
    }
    for name, (F, H, W, KP, late) in cases.items():
        ids, feats, n_ids, scene = feature_stream(F, H, W, KP)
        if late:                                                 # one empty message as well (:148-149 with no feature)
            ids = ids.copy(); ids[3] = -1
        slam, inputs = drive_reference(ref, shim, ids, feats, scene, tf_missing_first=late)
        out = capture(ref, shim, slam, ids, feats, inputs)
        out["n_ids"] = np.array(n_ids)
        out["imu"] = scene["imu"][:, :, :6]
        out["readme"] = np.array(README)
        path = os.path.join(HERE, f"ref_batch_{name}.npz")
        np.savez_compressed(path, **out)
        print(f"{path}: {len(out['lm_id'])} features, {len(out['stereo_meas'])} stereo factors, "
              f"{len(out['value_lm_key'])} landmarks, {os.path.getsize(path) // 1024} KiB, mse {float(out['report_mse']):.6g}")


if __name__ == "__main__":
    main()
