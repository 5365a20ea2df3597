"""Committed golden vectors (tests/golden/, made by tests/golden/make_golden.py from the oracle):
the oracle must keep reproducing them (CPU), and the HIP kernels must hit them on the GPU box."""
import os

import numpy as np
import pytest
import torch

from visual_underwater_slam_amd import ba_pack

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_oracle_reproduces_frontend_golden(oracle):
    g = np.load(os.path.join(G, "frontend_96x128.npz"))
    assert np.array_equal(oracle.fast_score(g["img"], 10), g["score"])
    keys, cnt, blur = oracle.fast_detect(g["img"], thr=10, border=20, cand_cap=2048)
    assert np.array_equal(cnt, g["cand_count"]) and np.array_equal(blur, g["blur"])
    assert np.array_equal(np.stack([np.sort(keys[n]) for n in range(2)]), g["cand_sorted"])
    kp, kc = oracle.select_topk(keys, cnt, 64)
    assert np.array_equal(kp, g["kp_keys"]) and np.array_equal(kc, g["kp_count"])
    desc, ang = oracle.orient_rbrief(g["img"], blur, kp, kc)
    assert np.array_equal(desc, g["desc"]) and np.array_equal(ang, g["angle"])
    idx, dist = oracle.hamming_match(desc, kp, kc, 128, [0, 0], [1, 0], max_dy=5, min_disp=0, max_disp=64, max_dist=80)
    assert np.array_equal(idx, g["match_idx"]) and np.array_equal(dist, g["match_dist"])
    assert (idx[1, :kc[0]] == np.arange(kc[0])).all()        # self-match row


def _pyramid_chain(oracle, g):
    from visual_underwater_slam_amd.frontend import pyramid_layout
    sizes, quotas = pyramid_layout(96, 128, 64, 3, 1.2)
    assert np.array_equal(np.array(sizes), g["sizes"]) and np.array_equal(np.array(quotas), g["quotas"])
    m = oracle.new_merged(2, 64)
    lvl, imgs = g["img"], []
    for l, ((h, w), q) in enumerate(zip(sizes, quotas)):
        if l > 0:
            lvl = oracle.resize_bilinear(lvl, h, w)
            imgs.append(lvl)
        ck, cc, blur = oracle.fast_detect(lvl, thr=10, border=20, cand_cap=2048)
        kp, kc = oracle.select_topk(ck, cc, q)
        desc, ang = oracle.orient_rbrief(lvl, blur, kp, kc)
        oracle.pyramid_append(kp, kc, desc, ang, h, w, l, 96, 128, m)
    return m, imgs


def _same_merged(got, g):
    c = g["kp_count"]
    assert np.array_equal(got["kp_count"], c) and np.array_equal(got["kp_keys"], g["kp_keys"]) and c.min() > 20
    for n in range(2):
        for k in ("desc", "angle", "kp_level", "kp_xy_q4"):
            assert np.array_equal(got[k][n, :c[n]], g[k][n, :c[n]]), k


def test_oracle_reproduces_pyramid_golden(oracle):
    g = np.load(os.path.join(G, "pyramid_96x128.npz"))
    m, imgs = _pyramid_chain(oracle, g)
    assert np.array_equal(imgs[0], g["level1"]) and np.array_equal(imgs[1], g["level2"])
    _same_merged(m, g)
    assert set(np.unique(g["kp_level"][0, :g["kp_count"][0]])) == {0, 1, 2}


def test_oracle_reproduces_ba_golden(oracle):
    g = np.load(os.path.join(G, "ba_c1.npz"))
    for i in range(8):
        r, H1, H2 = oracle.stereo_factor(g["f_T"][i], g["f_p"][i], g["f_m"][i], g["K"], 0.1)
        assert np.allclose(r, g["f_r"][i], rtol=1e-12) and np.allclose(H1, g["f_H1"][i], rtol=1e-12, atol=1e-12)
        assert np.allclose(H2, g["f_H2"][i], rtol=1e-12, atol=1e-12)
    nL = len(g["points_init"])
    pk = ba_pack.pack_observations(torch.from_numpy(g["obs_pose"]), torch.from_numpy(g["obs_point"]),
                                   torch.from_numpy(g["meas"]), 50, nL)
    st = ba_pack.build_structure(pk)
    P = oracle.BAProblem(pk, g["seq_K"], float(g["sigma"]), (np.array([0], np.int32), g["prior_T"], g["prior_sigmas"][None]))
    poses, points, rep = oracle.ba_lm_optimize(P, st["band"], g["poses_init"], g["points_init"])
    assert [rep["iterations"], rep["outer"], rep["tries"], rep["status"]] == g["counts"].tolist()
    assert np.allclose([rep["initial_error"]] + rep["err_hist"], g["err_hist"], rtol=1e-9)
    assert np.allclose(poses, g["poses_opt"], rtol=1e-8, atol=1e-10) and np.allclose(points, g["points_opt"], rtol=1e-8, atol=1e-9)


@pytest.mark.gpu
def test_hip_frontend_hits_golden(gpu):
    from visual_underwater_slam_amd.frontend import StereoOrbFrontend, ImageProcessorParams
    g = np.load(os.path.join(G, "frontend_96x128.npz"))
    prm = ImageProcessorParams(max_features=64, border=20, cand_cap=2048, max_disparity=64, stereo_max_distance=80)
    fe = StereoOrbFrontend(96, 128, max_frames=1, params=prm)
    res = fe.process(torch.from_numpy(g["img"][None]).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(res.kp_keys.cpu().numpy().view(np.uint32), g["kp_keys"])
    assert np.array_equal(res.desc.cpu().numpy().view(np.uint64), g["desc"])
    assert np.array_equal(res.angle.cpu().numpy(), g["angle"])
    assert np.array_equal(fe.blur.cpu().numpy(), g["blur"])
    assert np.array_equal(res.stereo_idx.cpu().numpy()[0], g["match_idx"][0])
    assert np.array_equal(res.stereo_dist.cpu().numpy()[0], g["match_dist"][0])


@pytest.mark.gpu
def test_hip_pyramid_hits_golden(gpu):
    from visual_underwater_slam_amd.frontend import StereoOrbFrontend, ImageProcessorParams
    g = np.load(os.path.join(G, "pyramid_96x128.npz"))
    prm = ImageProcessorParams(max_features=64, border=20, cand_cap=2048, n_levels=3)
    fe = StereoOrbFrontend(96, 128, max_frames=1, params=prm)
    res = fe.process(torch.from_numpy(g["img"][None]).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(fe.levels[1]["img"].cpu().numpy(), g["level1"])
    assert np.array_equal(fe.levels[2]["img"].cpu().numpy(), g["level2"])
    got = dict(kp_keys=res.kp_keys.cpu().numpy().view(np.uint32), kp_count=res.kp_count.cpu().numpy(),
               desc=res.desc.cpu().numpy().view(np.uint64), angle=res.angle.cpu().numpy(),
               kp_level=res.kp_level.cpu().numpy(), kp_xy_q4=res.kp_xy_q4.cpu().numpy())
    _same_merged(got, g)


@pytest.mark.gpu
def test_hip_ba_hits_golden(gpu):
    from visual_underwater_slam_amd.ba import StereoBAProblem, StereoBASolver
    g = np.load(os.path.join(G, "ba_c1.npz"))
    nL = len(g["points_init"])
    prob = StereoBAProblem(g["obs_pose"], g["obs_point"], g["meas"], 50, nL, g["seq_K"], float(g["sigma"]),
                           prior_pose=[0], prior_T=g["prior_T"], prior_sigmas=g["prior_sigmas"][None])
    poses, points, rep = StereoBASolver(prob).optimize(torch.from_numpy(g["poses_init"]).cuda(),
                                                      torch.from_numpy(g["points_init"]).cuda())
    assert [rep.iterations, rep.outer, rep.tries, rep.status] == g["counts"].tolist()
    assert np.allclose([rep.initial_error] + rep.err_hist, g["err_hist"], rtol=1e-8)
    rel = np.abs(poses.cpu().numpy() - g["poses_opt"]).max() / np.abs(g["poses_opt"]).max()
    assert rel < 1e-6                                           # north_star tolerance: 1e-4 relative
    assert np.abs(points.cpu().numpy() - g["points_opt"]).max() / np.abs(g["points_opt"]).max() < 1e-6
