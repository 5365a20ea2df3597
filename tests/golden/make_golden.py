#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle (run from the repo root: python tests/golden/make_golden.py).

The reference ships no golden vectors for this path and neither OpenCV nor GTSAM can be
run here, so these fixtures pin the oracle's own outputs (regression pin for the oracle, parity target
for the HIP kernels on the GPU box).  Fixtures are data only: inputs and expected outputs."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from visual_underwater_slam_amd import synth, ba_pack  # noqa: E402
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def frontend():
    img = synth.stereo_frames(42, 1, H=96, W=128)[0]                 # 2 images 96x128
    score = O.fast_score(img, 10)
    keys, cnt, blur = O.fast_detect(img, thr=10, border=20, cand_cap=2048)
    kp, kc = O.select_topk(keys, cnt, 64)
    desc, ang = O.orient_rbrief(img, blur, kp, kc)
    idx, dist = O.hamming_match(desc, kp, kc, 128, [0, 0], [1, 0], max_dy=5, min_disp=0, max_disp=64, max_dist=80)
    np.savez_compressed(os.path.join(HERE, "frontend_96x128.npz"), img=img, score=score, blur=blur,
                        cand_sorted=np.stack([np.sort(keys[n]) for n in range(2)]), cand_count=cnt,
                        kp_keys=kp, kp_count=kc, desc=desc, angle=ang, match_idx=idx, match_dist=dist)


def pyramid():
    from visual_underwater_slam_amd.frontend import pyramid_layout
    img = synth.stereo_frames(42, 1, H=96, W=128)[0]
    sizes, quotas = pyramid_layout(96, 128, 64, 3, 1.2)
    m = O.new_merged(2, 64)
    lvl, lv_imgs = img, []
    for l, ((h, w), q) in enumerate(zip(sizes, quotas)):
        if l > 0:
            lvl = O.resize_bilinear(lvl, h, w)
            lv_imgs.append(lvl)
        ck, cc, blur = O.fast_detect(lvl, thr=10, border=20, cand_cap=2048)
        kp, kc = O.select_topk(ck, cc, q)
        desc, ang = O.orient_rbrief(lvl, blur, kp, kc)
        O.pyramid_append(kp, kc, desc, ang, h, w, l, 96, 128, m)
    c = m["kp_count"]
    for n in range(2):           # slots past the count are unspecified: zero them in the fixture
        m["desc"][n, c[n]:] = 0; m["angle"][n, c[n]:] = 0; m["kp_level"][n, c[n]:] = 0; m["kp_xy_q4"][n, c[n]:] = 0
    np.savez_compressed(os.path.join(HERE, "pyramid_96x128.npz"), img=img, level1=lv_imgs[0], level2=lv_imgs[1],
                        sizes=np.array(sizes), quotas=np.array(quotas), **m)


def ba():
    rng = np.random.default_rng(20261004)
    K = np.array([1827.0, 1827.5999755859375, 0.0, 968.9000244140625, 561.4000244140625, 0.063])
    T, p, m, r, H1, H2 = [], [], [], [], [], []
    for _ in range(8):
        A = rng.normal(size=(3, 3)); Q, _ = np.linalg.qr(A)
        if np.linalg.det(Q) < 0:
            Q[:, 0] *= -1
        t = rng.normal(size=3)
        q = np.array([rng.uniform(-1, 1), rng.uniform(-.6, .6), rng.uniform(1.5, 6)])
        Ti = np.concatenate([Q.reshape(-1), t]); pi = Q @ q + t; mi = rng.uniform(0, 1000, 3)
        ri, h1, h2 = O.stereo_factor(Ti, pi, mi, K, 0.1)
        T.append(Ti); p.append(pi); m.append(mi); r.append(ri); H1.append(h1); H2.append(h2)
    seq = synth.ba_sequence(50, 500, 100)                             # C1
    nL = len(seq["points_gt"])
    pk = ba_pack.pack_observations(torch.from_numpy(seq["obs_pose"]), torch.from_numpy(seq["obs_point"]),
                                   torch.from_numpy(seq["meas"]), 50, nL)
    st = ba_pack.build_structure(pk)
    P = O.BAProblem(pk, seq["K"], seq["sigma"], (np.array([0], np.int32), seq["poses_gt"][:1], seq["prior_sigmas"][None]))
    poses, points, rep = O.ba_lm_optimize(P, st["band"], seq["poses_init"], seq["points_init"])
    np.savez_compressed(os.path.join(HERE, "ba_c1.npz"), K=K, f_T=np.array(T), f_p=np.array(p), f_m=np.array(m),
                        f_r=np.array(r), f_H1=np.array(H1), f_H2=np.array(H2),
                        obs_pose=seq["obs_pose"], obs_point=seq["obs_point"], meas=seq["meas"],
                        poses_init=seq["poses_init"], points_init=seq["points_init"], prior_T=seq["poses_gt"][:1],
                        prior_sigmas=seq["prior_sigmas"], sigma=seq["sigma"], seq_K=seq["K"],
                        err_hist=np.array([rep["initial_error"]] + rep["err_hist"]),
                        lambda_hist=np.array(rep["lambda_hist"]), poses_opt=poses, points_opt=points,
                        counts=np.array([rep["iterations"], rep["outer"], rep["tries"], rep["status"]]))


if __name__ == "__main__":
    frontend()
    pyramid()
    ba()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
