"""The end-to-end chain's CPU side: the rendered scene, and the oracle twins of the graph-building stage pinned against a
line-by-line numpy restatement of the reference's Python loops (/root/reference/batch.py:144-176 get_landmarks,
:253-266 batch_update, :270-305 batch_create) -- what vus_emit_stereo_factors computes for all keyframes at once."""
import numpy as np
import pytest
import torch

from visual_underwater_slam_amd import synth


def reference_loops(ids, feats, Rt, baseline, first_frame=1):
    """batch.py's per-message / per-landmark loops on plain Python containers.  Numbers follow :110-117, statements
    :152-166 and :295-305."""
    intrinsic = [1827.0, 1827.5999755859375, 968.9000244140625, 561.4000244140625]
    f = (intrinsic[0] + intrinsic[1]) / 2.0
    cx, cy = intrinsic[2], intrinsic[3]
    resolution_x, resolution_y = 1920, 1080
    landmark_accum = []
    for i in range(ids.shape[0]):                                   # one CameraMeasurement per keyframe
        R, t = Rt[i, :9].reshape(3, 3), Rt[i, 9:]
        landmarks = []
        for k in np.nonzero(ids[i] >= 0)[0]:                        # data.features in message order
            u0, v0, u1, v1 = feats[i, k]
            uL = (u0 + 1) * 0.5 * resolution_x
            uR = (u1 + 1) * 0.5 * resolution_x
            v = ((v0 + v1) / 2.0 + 1) * 0.5 * resolution_y
            d = uR - uL
            W = d / baseline
            cam_point = np.array([[(uL - cx) / W], [(v - cy) / W], [f / W]])
            world_point = R @ cam_point + t.reshape(3, 1)
            landmarks.append({'id': int(ids[i, k]), 'pose': world_point.reshape((3,)), 'uL': uL, 'uR': uR, 'v': v})
        landmark_accum.append(landmarks)
    initial, factors = {}, []
    for i in range(len(landmark_accum)):
        if i < first_frame:                                         # batch.py:280-305: no landmark loop for i == 0
            continue
        for lm in landmark_accum[i]:
            if lm['id'] not in initial:
                initial[lm['id']] = lm['pose']
            factors.append((i, lm['id'], lm['uL'], lm['uR'], lm['v']))
    return initial, factors


def synthetic_tracks(F=6, K=40, seed=3):
    rng = np.random.default_rng(seed)
    ids = -np.ones((F, K), np.int64)
    feats = np.zeros((F, K, 4))
    nxt = 0
    for f in range(F):
        for k in range(K):
            r = rng.random()
            if r < 0.35:
                continue
            if r < 0.7 and f > 0 and (ids[f - 1] >= 0).any():       # continue a track of the previous frame
                cand = ids[f - 1][ids[f - 1] >= 0]
                ids[f, k] = cand[rng.integers(len(cand))]
                if (ids[f, :k] == ids[f, k]).any():
                    ids[f, k] = nxt; nxt += 1
            else:
                ids[f, k] = nxt; nxt += 1
            u0 = rng.uniform(-0.9, 0.9)
            feats[f, k] = (u0, rng.uniform(-0.9, 0.9), u0 - rng.uniform(0.01, 0.05), rng.uniform(-0.9, 0.9))
    return ids, feats, nxt


@pytest.mark.parametrize("baseline", [0.063, -0.063])
def test_emit_stereo_factors_oracle_equals_the_reference_loops(oracle, baseline):
    ids, feats, n_ids = synthetic_tracks()
    s = synth.nav_sequence(6, 50, 5)
    cam = np.array([*synth.INTRINSIC, baseline, 1920, 1080, 0.0])
    of, oi, om, first, pt = oracle.emit_stereo_factors(ids, feats, s["poses_init"], cam, n_ids)
    initial, factors = reference_loops(ids, feats, s["poses_init"], baseline)
    assert len(factors) == len(of) > 50
    assert [(int(a), int(b)) for a, b in zip(of, oi)] == [(f[0], f[1]) for f in factors]     # batch_create's push order
    assert np.array_equal(om, np.array([f[2:] for f in factors]))                              # bit for bit
    seen = np.nonzero(first >= 0)[0]
    assert sorted(initial) == seen.tolist()
    for j in seen:
        assert np.allclose(pt[j], initial[j], rtol=0, atol=1e-12)      # R @ cam_point: BLAS summation order may differ
    K = ids.shape[1]
    for j in seen:                                                  # first sighting = first (frame >= 1, slot) carrying the id
        f, k = divmod(int(first[j]), K)
        assert f >= 1 and ids[f, k] == j and not (ids[1:f] == j).any() and not (ids[f, :k] == j).any()
    only0 = set(ids[0][ids[0] >= 0].tolist()) - set(ids[1:][ids[1:] >= 0].tolist())
    assert only0 and all(first[j] == -1 for j in only0)             # seen by keyframe 0 alone: never enters the graph


def test_reference_disparity_sign_puts_every_landmark_behind_its_camera(oracle):
    """batch.py:156 `d = uR - uL` with cam0 = left: z_cam = f * baseline / d < 0 (cheirality)."""
    ids, feats, n_ids = synthetic_tracks()
    s = synth.nav_sequence(6, 50, 5)
    K6 = np.array([synth.INTRINSIC[0], synth.INTRINSIC[1], 0.0, synth.INTRINSIC[2], synth.INTRINSIC[3], 0.063])
    for baseline, behind in ((0.063, True), (-0.063, False)):
        cam = np.array([*synth.INTRINSIC, baseline, 1920, 1080, 0.0])
        of, oi, om, first, pt = oracle.emit_stereo_factors(ids, feats, s["poses_init"], cam, n_ids)
        r0 = oracle.stereo_initial_residuals(s["poses_init"], K6, pt, of, oi, om)
        k = ids.shape[1]
        is_first = first[oi] == (of.astype(np.int64) * k + np.array([np.nonzero(ids[f] == j)[0][0] for f, j in zip(of, oi)]))
        assert is_first.sum() > 20
        if behind:
            assert np.isinf(r0[is_first]).all()                     # its own first sighting: behind the camera
        else:
            # ... or reprojects onto its own measurement, up to get_landmarks' f = (fx + fy) / 2 for all three
            # coordinates (batch.py:112,160-162) against the factor's separate fx, fy: < 0.2 px at the image border
            assert np.abs(r0[is_first]).max() < 0.2


def test_rendered_scene_is_identical_under_numpy_and_torch_and_has_the_plane_geometry(oracle):
    s = synth.scene_sequence(3, 240, 320, render=False)
    a = synth.scene_frames(s["poses_gt"], 240, 320)
    b = synth.scene_frames(s["poses_gt"], 240, 320, xp=torch)
    assert a.dtype == np.uint8 and a.shape == (3, 2, 240, 320) and np.array_equal(a, b.numpy())
    assert 100 < a.mean() < 156 and a.std() > 50                    # random 8-px blocks
    # the disparity of the rendered pair is the plane's: fx * b / depth at the image's pixel scale
    left, right = a[2, 0].astype(int), a[2, 1].astype(int)
    depth = s["poses_gt"][2, 11] - synth.SCENE_PLANE_Z
    want = synth.INTRINSIC[0] * synth.BASELINE_M / depth * 320 / 1920
    best = min(range(0, 12), key=lambda d: np.abs(left[60:180, 40 + d:280 + d] - right[60:180, 40:280]).mean())
    assert abs(best - want) <= 1.0
