"""numpy/ctypes wrapper of the CPU oracle (oracle/libvus_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never by the product package.  Same entry points as include/vus.h with a `_cpu` suffix,
host (numpy) buffers, synchronous.
"""
import ctypes
import os
import subprocess
from ctypes import c_int, c_void_p, c_double

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvus_oracle.so")
_lib = None


_native = None


def build(target="all", *make_vars):
    subprocess.check_call(["make", "-C", _HERE, "-s", target, *make_vars])


def _cpu_tag():
    """Short hash of this machine's CPU model and ISA flags: a -march=native library is only valid where it was built."""
    import hashlib
    txt = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith(("model name", "flags")):
                txt += line
                if line.startswith("flags"):
                    break
    except OSError:
        pass
    return hashlib.sha1(txt.encode()).hexdigest()[:10]


def lib(native=False):
    """The checker library; native=True: the -O3 -march=native build made ON this machine for timing only
    (bench.py's cpu_baseline leg).  VUS_ORACLE_LIB overrides the checker's path (sanitizer build)."""
    global _lib, _native
    if native:
        if _native is None:
            name = f"libvus_oracle_native_{_cpu_tag()}.so"
            build("native", f"NATIVE_LIB={name}")
            _native = ctypes.CDLL(os.path.join(_HERE, name))
        return _native
    if _lib is None:
        path = os.environ.get("VUS_ORACLE_LIB", LIB_PATH)
        if not os.path.exists(path):
            build()
        _lib = ctypes.CDLL(path)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


def _check(rc, name):
    if rc != 0:
        raise RuntimeError(f"oracle {name} failed ({rc})")


def _img_args(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    assert img.ndim == 3
    n, H, W = img.shape
    return img, n, H, W


def fast_score(img, thr=10):
    img, n, H, W = _img_args(img)
    out = np.empty((n, H, W), np.uint8)
    _check(lib().vus_fast_score_cpu(_p(img), n, H, W, W, int(thr), _p(out)), "fast_score")
    return out


def blur7(img):
    img, n, H, W = _img_args(img)
    out = np.empty((n, H, W), np.uint8)
    _check(lib().vus_blur7_cpu(_p(img), n, H, W, W, _p(out)), "blur7")
    return out


def fast_detect(img, thr=10, border=31, cand_cap=32768, want_blur=True, _lib=None):
    img, n, H, W = _img_args(img)
    keys = np.full((n, cand_cap), 0xFFFFFFFF, np.uint32)
    cnt = np.zeros(n, np.int32)
    blur = np.empty((n, H, W), np.uint8) if want_blur else None
    _check((_lib or lib()).vus_fast_detect_cpu(_p(img), n, H, W, W, int(thr), int(border), _p(blur), _p(keys),
                                     int(cand_cap), _p(cnt)), "fast_detect")
    return keys, cnt, blur


def fast_threshold_estimate(img, thr=10, border=31, max_kp=2000, sample_stride=16):
    """(hist [n,256], thr_img [n]) of the adaptive detector's sampling step."""
    img, n, H, W = _img_args(img)
    hist = np.zeros((n, 256), np.int32); thr_img = np.zeros(n, np.int32)
    _check(lib().vus_fast_threshold_estimate_cpu(_p(img), n, H, W, W, int(thr), int(border), int(max_kp), int(sample_stride),
                                                 _p(hist), _p(thr_img)), "fast_threshold_estimate")
    return hist, thr_img


def fast_detect_adaptive(img, thr_img, thr=10, border=31, max_kp=2000, cand_cap=32768):
    """Adaptive pass + the retry check: (keys, count, retried images)."""
    img, n, H, W = _img_args(img)
    thr_img = np.ascontiguousarray(thr_img, np.int32)
    keys = np.full((n, cand_cap), 0xFFFFFFFF, np.uint32); cnt = np.zeros(n, np.int32)
    _check(lib().vus_fast_detect_adaptive_cpu(_p(img), n, H, W, W, _p(thr_img), int(border), None, _p(keys), int(cand_cap), _p(cnt)),
           "fast_detect_adaptive")
    lst = np.zeros(n, np.int32); m = np.zeros(1, np.int32)
    _check(lib().vus_fast_detect_retry_cpu(_p(img), n, H, W, W, int(thr), _p(thr_img), int(max_kp), int(border), _p(keys),
                                           int(cand_cap), _p(cnt), _p(lst), _p(m)), "fast_detect_retry")
    return keys, cnt, lst[:int(m[0])]


def select_topk(cand_keys, cand_count, max_kp, _lib=None):
    cand_keys = np.ascontiguousarray(cand_keys, np.uint32)
    cand_count = np.ascontiguousarray(cand_count, np.int32)
    n, cap = cand_keys.shape
    kp = np.empty((n, max_kp), np.uint32)
    cnt = np.empty(n, np.int32)
    _check((_lib or lib()).vus_select_topk_cpu(_p(cand_keys), _p(cand_count), n, cap, int(max_kp), _p(kp), _p(cnt)),
           "select_topk")
    return kp, cnt


def select_grid(cand_keys, cand_count, H, W, grid_row, grid_col, per_cell, max_kp):
    cand_keys = np.ascontiguousarray(cand_keys, np.uint32)
    cand_count = np.ascontiguousarray(cand_count, np.int32)
    n, cap = cand_keys.shape
    kp = np.empty((n, max_kp), np.uint32)
    cnt = np.empty(n, np.int32)
    _check(lib().vus_select_grid_cpu(_p(cand_keys), _p(cand_count), n, cap, int(H), int(W), int(grid_row), int(grid_col),
                                     int(per_cell), int(max_kp), _p(kp), _p(cnt)), "select_grid")
    return kp, cnt


def orient_rbrief(img, blur, kp_keys, kp_count, _lib=None):
    img, n, H, W = _img_args(img)
    blur = np.ascontiguousarray(blur, np.uint8)
    kp_keys = np.ascontiguousarray(kp_keys, np.uint32)
    kp_count = np.ascontiguousarray(kp_count, np.int32)
    max_kp = kp_keys.shape[1]
    desc = np.empty((n, max_kp, 4), np.uint64)
    ang = np.empty((n, max_kp), np.uint8)
    _check((_lib or lib()).vus_orient_rbrief_cpu(_p(img), _p(blur), n, H, W, W, _p(kp_keys), _p(kp_count), max_kp,
                                       _p(desc), _p(ang)), "orient_rbrief")
    return desc, ang


def steering_tables(which=0):
    """(rot int8 [30,256,4], cos_q14 int32 [30], sin_q14 int32 [30]).  which=0: derived by the oracle itself with
    libm at load time (what vus_orient_rbrief_cpu uses); which=1: the generated header include/vus_orb_tables.h
    (what the HIP kernels compile in)."""
    rot = np.zeros((30, 256, 4), np.int8); c = np.zeros(30, np.int32); sn = np.zeros(30, np.int32)
    _check(lib().vus_oracle_tables_cpu(int(which), _p(rot), _p(c), _p(sn)), "oracle_tables")
    return rot, c, sn


def resize_bilinear(img, Hd, Wd):
    img, n, H, W = _img_args(img)
    out = np.empty((n, Hd, Wd), np.uint8)
    _check(lib().vus_resize_bilinear_cpu(_p(img), n, H, W, W, _p(out), int(Hd), int(Wd), int(Wd)), "resize_bilinear")
    return out


def pyramid_append(lvl_keys, lvl_count, lvl_desc, lvl_angle, Hl, Wl, level, H0, W0, merged):
    """merged = dict(kp_keys, kp_count, desc, angle, kp_level, kp_xy_q4) of numpy arrays, updated in place."""
    lvl_keys = np.ascontiguousarray(lvl_keys, np.uint32)
    lvl_count = np.ascontiguousarray(lvl_count, np.int32)
    lvl_desc = np.ascontiguousarray(lvl_desc, np.uint64)
    lvl_angle = np.ascontiguousarray(lvl_angle, np.uint8)
    n, lk = lvl_keys.shape
    m = merged
    _check(lib().vus_pyramid_append_cpu(_p(lvl_keys), _p(lvl_count), _p(lvl_desc), _p(lvl_angle), n, lk, int(Hl), int(Wl),
                                        int(level), int(H0), int(W0), m["kp_keys"].shape[1], _p(m["kp_keys"]),
                                        _p(m["kp_count"]), _p(m["desc"]), _p(m["angle"]), _p(m["kp_level"]),
                                        _p(m["kp_xy_q4"])), "pyramid_append")
    return m


def new_merged(n_img, max_kp):
    return dict(kp_keys=np.full((n_img, max_kp), 0xFFFFFFFF, np.uint32), kp_count=np.zeros(n_img, np.int32),
                desc=np.zeros((n_img, max_kp, 4), np.uint64), angle=np.zeros((n_img, max_kp), np.uint8),
                kp_level=np.zeros((n_img, max_kp), np.uint8), kp_xy_q4=np.zeros((n_img, max_kp, 2), np.int32))


def hamming_match(desc, kp_keys, kp_count, W, q_index, t_index, max_dy=-1, min_disp=0, max_disp=0,
                  max_dist=256, H=None, _lib=None):
    desc = np.ascontiguousarray(desc, np.uint64)
    kp_keys = np.ascontiguousarray(kp_keys, np.uint32)
    kp_count = np.ascontiguousarray(kp_count, np.int32)
    q_index = np.ascontiguousarray(q_index, np.int32)
    t_index = np.ascontiguousarray(t_index, np.int32)
    max_kp = kp_keys.shape[1]
    npairs = q_index.shape[0]
    idx = np.empty((npairs, max_kp), np.int32)
    dist = np.empty((npairs, max_kp), np.int32)
    if H is None:   # any bound on the row index will do for the oracle
        H = int((kp_keys & 0xFFFFFF).max()) // int(W) + 1 if kp_keys.size else 1
    _check((_lib or lib()).vus_hamming_match_cpu(_p(desc), _p(kp_keys), _p(kp_count), max_kp, int(H), int(W), _p(q_index),
                                       _p(t_index), npairs, int(max_dy), int(min_disp), int(max_disp),
                                       int(max_dist), _p(idx), _p(dist)), "hamming_match")
    return idx, dist


def cross_check(idx_fwd, idx_bwd):
    idx_fwd = np.ascontiguousarray(idx_fwd, np.int32)
    idx_bwd = np.ascontiguousarray(idx_bwd, np.int32)
    out = np.empty_like(idx_fwd)
    _check(lib().vus_cross_check_cpu(_p(idx_fwd), _p(idx_bwd), idx_fwd.shape[0], idx_fwd.shape[1], _p(out)), "cross_check")
    return out


def track_ids(stereo_idx, track_idx, kp_keys, kp_count, H, W):
    stereo_idx = np.ascontiguousarray(stereo_idx, np.int32)
    F, K = stereo_idx.shape
    track_idx = np.ascontiguousarray(track_idx, np.int32) if track_idx is not None and F > 1 else None
    kp_keys = np.ascontiguousarray(kp_keys, np.uint32)
    kp_count = np.ascontiguousarray(kp_count, np.int32)
    ids = np.empty((F, K), np.int64); feat = np.empty((F, K, 4), np.float64); n = np.zeros(1, np.int64)
    _check(lib().vus_track_ids_cpu(_p(stereo_idx), _p(track_idx), _p(kp_keys), _p(kp_count), F, K, int(H), int(W),
                                   _p(ids), _p(feat), _p(n)), "track_ids")
    return ids, feat, int(n[0])


def triangulate(feat, cam, Rt):
    feat = np.ascontiguousarray(feat, np.float64)
    cam = np.ascontiguousarray(cam, np.float64)
    Rt = np.ascontiguousarray(Rt, np.float64)
    n = feat.shape[0]
    out = np.empty((n, 6), np.float64)
    _check(lib().vus_triangulate_cpu(_p(feat), n, _p(cam), _p(Rt), _p(out)), "triangulate")
    return out


def emit_stereo_factors(ids, feat, Rt, cam, n_ids, first_frame=1):
    """batch_update + batch_create's landmark loop (batch.py:264-265, 295-305) over all keyframes: returns
    (obs_frame, obs_id, obs_meas, lm_first, lm_point)."""
    ids = np.ascontiguousarray(ids, np.int64)
    F, K = ids.shape
    feat = np.ascontiguousarray(feat, np.float64); Rt = np.ascontiguousarray(Rt, np.float64).reshape(F, 12)
    cam = np.ascontiguousarray(cam, np.float64)
    base = np.zeros(F + 1, np.int32); count = np.zeros(1, np.int32)
    of = np.zeros(F * K, np.int32); oi = np.zeros(F * K, np.int64); om = np.zeros((F * K, 3), np.float64)
    first = np.zeros(max(int(n_ids), 1), np.int64); pt = np.zeros((max(int(n_ids), 1), 3), np.float64)
    _check(lib().vus_emit_stereo_factors_cpu(_p(ids), _p(feat), _p(Rt), _p(cam), F, K, int(first_frame),
                                             ctypes.c_longlong(int(n_ids)), _p(base), _p(count), _p(of), _p(oi), _p(om),
                                             _p(first), _p(pt)), "emit_stereo_factors")
    n = int(count[0])
    return of[:n], oi[:n], om[:n], first[:int(n_ids)], pt[:int(n_ids)]


def stereo_initial_residuals(Rt, K, lm_point, obs_frame, obs_id, obs_meas):
    Rt = np.ascontiguousarray(Rt, np.float64); K = np.ascontiguousarray(K, np.float64)
    lm_point = np.ascontiguousarray(lm_point, np.float64)
    obs_frame = np.ascontiguousarray(obs_frame, np.int32); obs_id = np.ascontiguousarray(obs_id, np.int64)
    obs_meas = np.ascontiguousarray(obs_meas, np.float64)
    n = len(obs_frame)
    out = np.empty((n, 3), np.float64)
    _check(lib().vus_stereo_initial_residuals_cpu(_p(Rt), _p(K), _p(lm_point), _p(obs_frame), _p(obs_id), _p(obs_meas), n,
                                                  _p(out)), "stereo_initial_residuals")
    return out


# ---------------------------------------------------------------------------------------------
# bundle adjustment (oracle/vus_oracle_ba.c)
class _BAProblem(ctypes.Structure):
    _fields_ = [("n_poses", c_int), ("n_points", c_int), ("n_obs", c_int), ("n_priors", c_int),
                ("K", c_void_p), ("inv_sigma", c_double), ("meas", c_void_p), ("obs_pose", c_void_p),
                ("obs_point", c_void_p), ("point_ptr", c_void_p), ("obs_ppos", c_void_p),
                ("pose_ptr", c_void_p), ("pobs_lidx", c_void_p), ("prior_pose", c_void_p),
                ("prior_T", c_void_p), ("prior_w", c_void_p), ("pose_stride", c_int)]


class _BAStructure(ctypes.Structure):
    _fields_ = [("band", c_int), ("n_blocks", c_int), ("n_pairs", c_int), ("blk_ptr", c_void_p),
                ("blk_i", c_void_p), ("blk_k", c_void_p), ("pair_a", c_void_p), ("pair_b", c_void_p)]


class _LMParams(ctypes.Structure):
    _fields_ = [("lambda_initial", c_double), ("lambda_factor", c_double), ("lambda_upper", c_double),
                ("lambda_lower", c_double), ("min_model_fidelity", c_double), ("rel_tol", c_double),
                ("abs_tol", c_double), ("error_tol", c_double), ("max_iterations", c_int)]


class _LMReport(ctypes.Structure):
    _fields_ = [("iterations", c_int), ("outer", c_int), ("tries", c_int), ("status", c_int),
                ("initial_error", c_double), ("final_error", c_double), ("final_lambda", c_double),
                ("err_hist", c_double * 128), ("lambda_hist", c_double * 128)]


class BAProblem:
    """Host-side (numpy) vus_ba_problem.  `pk` is the dict of ba_pack.pack_observations (numpy or
    CPU torch tensors), priors = (pose_idx [n], T [n,12], sigmas [n,6])."""

    def __init__(self, pk, K, sigma, priors=None):
        def np_(x, dt):
            x = x.numpy() if hasattr(x, "numpy") else x
            return np.ascontiguousarray(x, dtype=dt)
        self.n_poses, self.n_points, self.n_obs = pk["n_poses"], pk["n_points"], pk["n_obs"]
        self.K = np_(K, np.float64)
        self.meas = np_(pk["meas"], np.float64)
        self.arr = {k: np_(pk[k], np.int32) for k in
                    ("obs_pose", "obs_point", "point_ptr", "obs_ppos", "pose_ptr", "pobs_lidx")}
        if priors is None:
            priors = (np.zeros(0, np.int32), np.zeros((0, 12)), np.zeros((0, 6)))
        self.prior_pose = np_(priors[0], np.int32)
        self.prior_T = np_(priors[1], np.float64)
        self.prior_w = np.ascontiguousarray(1.0 / np_(priors[2], np.float64)) if len(priors[0]) else np.zeros((0, 6))
        self.c = _BAProblem(self.n_poses, self.n_points, self.n_obs, len(self.prior_pose), _p(self.K).value,
                            1.0 / float(sigma), _p(self.meas).value, _p(self.arr["obs_pose"]).value,
                            _p(self.arr["obs_point"]).value, _p(self.arr["point_ptr"]).value,
                            _p(self.arr["obs_ppos"]).value, _p(self.arr["pose_ptr"]).value,
                            _p(self.arr["pobs_lidx"]).value, _p(self.prior_pose).value if len(self.prior_pose) else None,
                            _p(self.prior_T).value if len(self.prior_pose) else None,
                            _p(self.prior_w).value if len(self.prior_pose) else None, 1)

    def ref(self):
        return ctypes.byref(self.c)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def ba_error(P, poses, points):
    e = np.zeros(1)
    _check(lib().vus_ba_error_cpu(P.ref(), _p(_f64(poses)), _p(_f64(points)), _p(e)), "ba_error")
    return float(e[0])


def ba_linearize(P, poses, points):
    W = np.zeros((P.n_obs, 18)); V = np.zeros((P.n_points, 6)); gl = np.zeros((P.n_points, 3))
    Hpp = np.zeros((P.n_poses, 36)); gp = np.zeros((P.n_poses, 6)); e = np.zeros(1)
    _check(lib().vus_ba_linearize_cpu(P.ref(), _p(_f64(poses)), _p(_f64(points)), _p(W), _p(V), _p(gl),
                                      _p(Hpp), _p(gp), _p(e)), "ba_linearize")
    return {"W": W, "V": V, "gl": gl, "Hpp": Hpp, "gp": gp, "err": float(e[0])}


def ba_structure(P, band):
    """vus_ba_structure arrays (numpy dict, keys of ba_pack.build_structure) by the plain row-by-row statement."""
    nP = P.n_poses
    rb, rp = np.zeros(nP, np.int32), np.zeros(nP, np.int32)
    _check(lib().vus_ba_structure_count_cpu(P.ref(), int(band), _p(rb), _p(rp)), "ba_structure_count")
    bb = np.zeros(nP + 1, np.int32); bb[1:] = np.cumsum(rb)
    pb = np.zeros(nP + 1, np.int32); pb[1:] = np.cumsum(rp)
    nb, npair = int(bb[-1]), int(pb[-1])
    out = {"band": int(band), "n_blocks": nb, "n_pairs": npair, "blk_ptr": np.zeros(nb + 1, np.int32),
           "blk_i": np.zeros(nb, np.int32), "blk_k": np.zeros(nb, np.int32), "pair_a": np.zeros(npair, np.int32),
           "pair_b": np.zeros(npair, np.int32)}
    _check(lib().vus_ba_structure_fill_cpu(P.ref(), int(band), _p(bb), _p(pb), _p(out["blk_ptr"]), _p(out["blk_i"]),
                                           _p(out["blk_k"]), _p(out["pair_a"]), _p(out["pair_b"])), "ba_structure_fill")
    return out


def ba_tiles(P, band):
    """vus_ba_tiles arrays by the plain statement (oracle/vus_oracle_pack.c): dict with unit_ptr, entries [n,4], order."""
    cnt = np.zeros(max(P.n_points, 1), np.int32)
    _check(lib().vus_ba_tiles_count_cpu(P.ref(), _p(cnt)), "ba_tiles_count")
    base = np.zeros(P.n_points + 1, np.int32); base[1:] = np.cumsum(cnt[:P.n_points])
    n = int(base[-1])
    n_tiles, dt1 = (P.n_poses + 7) // 8, (int(band) + 7) // 8 + 1
    out = {"band": int(band), "n_tiles": n_tiles, "n_units": n_tiles * dt1, "n_entries": n, "lm_entries": cnt[:P.n_points],
           "unit_ptr": np.zeros(n_tiles * dt1 + 1, np.int32), "entries": np.zeros((max(n, 1), 4), np.int32),
           "order": np.zeros(n_tiles * dt1, np.int32)}
    _check(lib().vus_ba_tiles_fill_cpu(P.ref(), int(band), _p(base), n, _p(out["unit_ptr"]), _p(out["entries"]),
                                       _p(out["order"]), None, ctypes.c_longlong(0)), "ba_tiles_fill")
    out["entries"] = out["entries"][:n]
    return out


def ba_schur(P, band, lam, lin):
    """Twin of vus_ba_schur (W, Y in L-order; the tile lists are a schedule of the same sum and are not needed here)."""
    Vinv = np.zeros((P.n_points, 6)); Y = np.zeros((P.n_obs, 18))
    Sb = np.zeros((P.n_poses, band + 1, 36)); gs = np.zeros((P.n_poses, 6))
    _check(lib().vus_ba_schur_cpu(P.ref(), None, c_double(lam), _p(lin["W"]), _p(lin["V"]), _p(lin["gl"]), _p(lin["Hpp"]),
                                  _p(lin["gp"]), _p(Vinv), _p(Y), _p(Sb), int(band), _p(gs), None), "ba_schur")
    return {"Vinv": Vinv, "Y": Y, "Sband": Sb, "gs": gs}


def ba_band_solve(Sband, gs):
    Sb = np.array(Sband, dtype=np.float64, order="C", copy=True)
    nP, B1 = Sb.shape[0], Sb.shape[1]
    dp = np.zeros((nP, 6)); st = np.zeros(1, np.int32)
    _check(lib().vus_ba_band_solve_cpu(_p(Sb), nP, B1 - 1, _p(_f64(gs)), _p(dp), _p(st)), "ba_band_solve")
    return dp, int(st[0]), Sb


def ba_backsub(P, lin, Vinv, dp):
    dl = np.zeros((P.n_points, 3))
    _check(lib().vus_ba_backsub_cpu(P.ref(), _p(lin["W"]), _p(_f64(Vinv)), _p(lin["gl"]), _p(_f64(dp)), _p(dl)),
           "ba_backsub")
    return dl


def ba_eval_step(P, poses, points, dp, dl):
    npo = np.zeros((P.n_poses, 12)); npt = np.zeros((P.n_points, 3)); out = np.zeros(2)
    _check(lib().vus_ba_eval_step_cpu(P.ref(), _p(_f64(poses)), _p(_f64(points)), _p(_f64(dp)), _p(_f64(dl)),
                                      _p(npo), _p(npt), _p(out)), "ba_eval_step")
    return npo, npt, float(out[0]), float(out[1])


LM_DEFAULTS = dict(lambda_initial=1e-5, lambda_factor=10.0, lambda_upper=1e5, lambda_lower=0.0,
                   min_model_fidelity=1e-3, rel_tol=1e-5, abs_tol=1e-5, error_tol=0.0, max_iterations=100)


def ba_lm_optimize(P, band, poses, points, **params):
    prm = dict(LM_DEFAULTS); prm.update(params)
    c = _LMParams(*[prm[k] for k, _ in _LMParams._fields_])
    rep = _LMReport()
    poses = np.array(poses, dtype=np.float64, order="C", copy=True)
    points = np.array(points, dtype=np.float64, order="C", copy=True)
    _check(lib().vus_ba_lm_optimize_cpu(P.ref(), int(band), ctypes.byref(c), _p(poses), _p(points),
                                        ctypes.byref(rep)), "ba_lm_optimize")
    n = min(rep.outer, 128)
    return poses, points, {"iterations": rep.iterations, "outer": rep.outer, "tries": rep.tries,
                           "status": rep.status, "initial_error": rep.initial_error,
                           "final_error": rep.final_error, "final_lambda": rep.final_lambda,
                           "err_hist": list(rep.err_hist[:n]), "lambda_hist": list(rep.lambda_hist[:n])}


def stereo_factor(T, p, m, K, w):
    r = np.zeros(3); H1 = np.zeros(18); H2 = np.zeros(9)
    lib().vus_stereo_factor_cpu(_p(_f64(T)), _p(_f64(p)), _p(_f64(m)), _p(_f64(K)), c_double(w), _p(r), _p(H1), _p(H2))
    return r, H1.reshape(3, 6), H2.reshape(3, 3)


def pose_retract(T, xi):
    out = np.zeros(12)
    lib().vus_pose_retract_cpu(_p(_f64(T)), _p(_f64(xi)), _p(out))
    return out


def pose_local(T, T2):
    xi = np.zeros(6)
    lib().vus_pose_local_cpu(_p(_f64(T)), _p(_f64(T2)), _p(xi))
    return xi


# ---------------------------------------------------------------------------------------------
# inertial / velocity factors (oracle/vus_oracle_nav.c)
PIM = dict(DT=0, DR=1, DP=10, DV=13, DR_DBG=16, DP_DBA=25, DP_DBG=34, DV_DBA=43, DV_DBG=52, BIAS=61, COV=67, N=148)


def imu_preintegrate(samples, bias_hat, acc_cov, gyro_cov, int_cov):
    samples = _f64(np.asarray(samples).reshape(-1, 7))
    pim = np.zeros(PIM["N"])
    _check(lib().vus_imu_preintegrate_cpu(_p(samples), samples.shape[0], _p(_f64(bias_hat)), _p(_f64(acc_cov)),
                                          _p(_f64(gyro_cov)), _p(_f64(int_cov)), _p(pim)), "imu_preintegrate")
    return pim


def sqrt_information(cov):
    cov = _f64(cov)
    n = cov.shape[0]
    W = np.zeros((n, n))
    _check(lib().vus_sqrt_information_cpu(_p(cov), n, _p(W)), "sqrt_information")
    return W


def imu_factor(Ti, vi, Tj, vj, bias, pim, g, jac=True):
    r = np.zeros(9)
    J = np.zeros((9, 24)) if jac else None
    lib().vus_imu_factor_cpu(_p(_f64(Ti)), _p(_f64(vi)), _p(_f64(Tj)), _p(_f64(vj)), _p(_f64(bias)), _p(_f64(pim)),
                             _p(_f64(g)), _p(r), _p(J))
    return (r, J) if jac else r


def dvl_factor(T, v, m):
    e = np.zeros(3); JX = np.zeros((3, 6)); Jv = np.zeros((3, 3))
    lib().vus_dvl_factor_cpu(_p(_f64(T)), _p(_f64(v)), _p(_f64(m)), _p(e), _p(JX), _p(Jv))
    return e, JX, Jv


def so3_expmap(w):
    R = np.zeros(9)
    lib().vus_so3_expmap_cpu(_p(_f64(w)), _p(R))
    return R.reshape(3, 3)


class _NavFactors(ctypes.Structure):
    _fields_ = [("n_imu", c_int), ("imu_i", c_void_p), ("imu_j", c_void_p), ("imu_pim", c_void_p), ("imu_W", c_void_p),
                ("gravity", c_double * 3), ("n_dvl", c_int), ("dvl_pose", c_void_p), ("dvl_meas", c_void_p),
                ("dvl_w", c_void_p), ("n_vprior", c_int), ("vprior_idx", c_void_p), ("vprior_v", c_void_p),
                ("vprior_w", c_void_p)]


class NavFactors:
    """Host-side vus_nav_factors: imu = (i [n], j [n], pim [n,148], W [n,81]), dvl = (pose [n], meas [n,3], sigma [n]),
    vprior = (idx [n], v [n,3], sigmas [n,3])."""

    def __init__(self, gravity, imu=None, dvl=None, vprior=None):
        z = np.zeros(0)
        self.imu_i = np.ascontiguousarray(imu[0] if imu else z, np.int32)
        self.imu_j = np.ascontiguousarray(imu[1] if imu else z, np.int32)
        self.imu_pim = _f64(imu[2] if imu else z)
        self.imu_W = _f64(imu[3] if imu else z)
        self.dvl_pose = np.ascontiguousarray(dvl[0] if dvl else z, np.int32)
        self.dvl_meas = _f64(dvl[1] if dvl else z)
        self.dvl_w = _f64(1.0 / np.asarray(dvl[2], float)) if dvl else z
        self.vp_idx = np.ascontiguousarray(vprior[0] if vprior else z, np.int32)
        self.vp_v = _f64(vprior[1] if vprior else z)
        self.vp_w = _f64(1.0 / np.asarray(vprior[2], float)) if vprior else z
        pv = lambda a: _p(a).value if a.size else None
        self.c = _NavFactors(len(self.imu_i), pv(self.imu_i), pv(self.imu_j), pv(self.imu_pim), pv(self.imu_W),
                             (c_double * 3)(*[float(x) for x in gravity]), len(self.dvl_pose), pv(self.dvl_pose),
                             pv(self.dvl_meas), pv(self.dvl_w), len(self.vp_idx), pv(self.vp_idx), pv(self.vp_v), pv(self.vp_w))

    def ref(self):
        return ctypes.byref(self.c)


def nav_error(P, N, poses, vels, bias, points):
    e = np.zeros(1)
    _check(lib().vus_nav_total_error_cpu(P.ref(), N.ref(), _p(_f64(poses)), _p(_f64(vels)), _p(_f64(bias)), _p(_f64(points)), _p(e)),
           "nav_error")
    return float(e[0])


def nav_lm_optimize(P, N, poses, vels, bias, points, **params):
    prm = dict(LM_DEFAULTS); prm.update(params)
    c = _LMParams(*[prm[k] for k, _ in _LMParams._fields_])
    rep = _LMReport()
    cp = lambda a: np.array(a, dtype=np.float64, order="C", copy=True)
    poses, vels, bias, points = cp(poses), cp(vels), cp(bias), cp(points)
    _check(lib().vus_nav_lm_optimize_cpu(P.ref(), N.ref(), ctypes.byref(c), _p(poses), _p(vels), _p(bias), _p(points),
                                         ctypes.byref(rep)), "nav_lm_optimize")
    n = min(rep.outer, 128)
    return poses, vels, bias, points, {"iterations": rep.iterations, "outer": rep.outer, "tries": rep.tries,
                                       "status": rep.status, "initial_error": rep.initial_error,
                                       "final_error": rep.final_error, "final_lambda": rep.final_lambda,
                                       "err_hist": list(rep.err_hist[:n]), "lambda_hist": list(rep.lambda_hist[:n])}
