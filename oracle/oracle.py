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


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = ctypes.CDLL(LIB_PATH)
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


def fast_detect(img, thr=10, border=31, cand_cap=32768, want_blur=True):
    img, n, H, W = _img_args(img)
    keys = np.full((n, cand_cap), 0xFFFFFFFF, np.uint32)
    cnt = np.zeros(n, np.int32)
    blur = np.empty((n, H, W), np.uint8) if want_blur else None
    _check(lib().vus_fast_detect_cpu(_p(img), n, H, W, W, int(thr), int(border), _p(blur), _p(keys),
                                     int(cand_cap), _p(cnt)), "fast_detect")
    return keys, cnt, blur


def select_topk(cand_keys, cand_count, max_kp):
    cand_keys = np.ascontiguousarray(cand_keys, np.uint32)
    cand_count = np.ascontiguousarray(cand_count, np.int32)
    n, cap = cand_keys.shape
    kp = np.empty((n, max_kp), np.uint32)
    cnt = np.empty(n, np.int32)
    _check(lib().vus_select_topk_cpu(_p(cand_keys), _p(cand_count), n, cap, int(max_kp), _p(kp), _p(cnt)),
           "select_topk")
    return kp, cnt


def orient_rbrief(img, blur, kp_keys, kp_count):
    img, n, H, W = _img_args(img)
    blur = np.ascontiguousarray(blur, np.uint8)
    kp_keys = np.ascontiguousarray(kp_keys, np.uint32)
    kp_count = np.ascontiguousarray(kp_count, np.int32)
    max_kp = kp_keys.shape[1]
    desc = np.empty((n, max_kp, 4), np.uint64)
    ang = np.empty((n, max_kp), np.uint8)
    _check(lib().vus_orient_rbrief_cpu(_p(img), _p(blur), n, H, W, W, _p(kp_keys), _p(kp_count), max_kp,
                                       _p(desc), _p(ang)), "orient_rbrief")
    return desc, ang


def hamming_match(desc, kp_keys, kp_count, W, q_index, t_index, max_dy=-1, min_disp=0, max_disp=0,
                  max_dist=256):
    desc = np.ascontiguousarray(desc, np.uint64)
    kp_keys = np.ascontiguousarray(kp_keys, np.uint32)
    kp_count = np.ascontiguousarray(kp_count, np.int32)
    q_index = np.ascontiguousarray(q_index, np.int32)
    t_index = np.ascontiguousarray(t_index, np.int32)
    max_kp = kp_keys.shape[1]
    npairs = q_index.shape[0]
    idx = np.empty((npairs, max_kp), np.int32)
    dist = np.empty((npairs, max_kp), np.int32)
    _check(lib().vus_hamming_match_cpu(_p(desc), _p(kp_keys), _p(kp_count), max_kp, int(W), _p(q_index),
                                       _p(t_index), npairs, int(max_dy), int(min_disp), int(max_disp),
                                       int(max_dist), _p(idx), _p(dist)), "hamming_match")
    return idx, dist


def triangulate(feat, cam, Rt):
    feat = np.ascontiguousarray(feat, np.float64)
    cam = np.ascontiguousarray(cam, np.float64)
    Rt = np.ascontiguousarray(Rt, np.float64)
    n = feat.shape[0]
    out = np.empty((n, 6), np.float64)
    _check(lib().vus_triangulate_cpu(_p(feat), n, _p(cam), _p(Rt), _p(out)), "triangulate")
    return out
