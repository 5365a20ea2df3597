"""Differential fuzzing of the front-end against the oracle chain: random image sizes (odd widths, tiny
heights), thresholds, keypoint budgets, pyramid depths, cross-check on/off.  Every stage output must be bit-exact.
usage: python tools/fuzz_frontend.py [n_cases] [seed]"""
import sys
import numpy as np
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from visual_underwater_slam_amd import synth
from visual_underwater_slam_amd.frontend import StereoOrbFrontend, ImageProcessorParams, pyramid_layout
from oracle import oracle as O

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def oracle_chain(flat, p, H, W):
    sizes, quotas = pyramid_layout(H, W, p.max_features, p.n_levels, p.scale_factor)
    if p.n_levels == 1:
        ck, cc, blur = O.fast_detect(flat, p.fast_threshold, p.border, p.cand_cap)
        kp, kc = O.select_topk(ck, cc, p.max_features)
        desc, ang = O.orient_rbrief(flat, blur, kp, kc)
        return kp, kc, desc, ang
    m = O.new_merged(flat.shape[0], p.max_features)
    lvl = flat
    for l, ((h, w), q) in enumerate(zip(sizes, quotas)):
        if l > 0:
            lvl = O.resize_bilinear(lvl, h, w)
        ck, cc, blur = O.fast_detect(lvl, p.fast_threshold, p.border, p.cand_cap)
        kp, kc = O.select_topk(ck, cc, max(q, 1))
        desc, ang = O.orient_rbrief(lvl, blur, kp, kc)
        O.pyramid_append(kp, kc, desc, ang, h, w, l, H, W, m)
    return m["kp_keys"], m["kp_count"], m["desc"], m["angle"]


for case in range(n_cases):
    levels = int(rng.choice([1, 1, 2, 3]))
    border = int(rng.choice([16, 20, 31]))
    hmin = int(np.ceil((2 * border + 8) * 1.2 ** (levels - 1))) + 2
    H = int(rng.integers(max(hmin, 72), 260))
    W = int(rng.integers(max(hmin, 80), 420))
    F = int(rng.integers(1, 4))
    p = ImageProcessorParams(fast_threshold=int(rng.integers(5, 40)), max_features=int(rng.integers(20, 700)),
                             border=border, n_levels=levels, cross_check=bool(rng.integers(0, 2)),
                             stereo_threshold=int(rng.integers(1, 8)), max_disparity=int(rng.integers(20, 128)),
                             stereo_max_distance=int(rng.integers(30, 100)), track_max_distance=int(rng.integers(30, 100)))
    img = synth.stereo_frames(int(rng.integers(0, 500)), F, H=H, W=W)
    if rng.integers(0, 4) == 0:
        img[:, :, : H // 2] = 77              # half of the image flat: few keypoints, counts < budget
    fe = StereoOrbFrontend(H, W, max_frames=F, params=p)
    res = fe.process(torch.from_numpy(img).cuda())
    torch.cuda.synchronize()
    kp, kc, desc, ang = oracle_chain(img.reshape(2 * F, H, W), fe.p, H, W)
    tag = f"case {case}: {W}x{H} F={F} levels={levels} K={p.max_features} thr={p.fast_threshold} border={border} xc={p.cross_check}"
    assert np.array_equal(res.kp_count.cpu().numpy(), kc), tag
    assert np.array_equal(res.kp_keys.cpu().numpy().view(np.uint32), kp), tag
    gd, ga = res.desc.cpu().numpy().view(np.uint64), res.angle.cpu().numpy()
    for n in range(2 * F):
        assert np.array_equal(gd[n, :kc[n]], desc[n, :kc[n]]) and np.array_equal(ga[n, :kc[n]], ang[n, :kc[n]]), tag
    f = np.arange(F, dtype=np.int32)
    d2 = desc.copy()
    sidx, sdist = O.hamming_match(d2, kp, kc, W, 2 * f, 2 * f + 1, p.stereo_threshold, p.min_disparity, p.max_disparity,
                                  p.stereo_max_distance, H=H)
    if p.cross_check:
        bwd, _ = O.hamming_match(d2, kp, kc, W, 2 * f + 1, 2 * f, p.stereo_threshold, -p.max_disparity, -p.min_disparity,
                                 p.stereo_max_distance, H=H)
        sidx = O.cross_check(sidx, bwd)
    gi = res.stereo_idx.cpu().numpy()
    for r in range(F):
        assert np.array_equal(gi[r, :kc[2 * r]], sidx[r, :kc[2 * r]]), tag + " stereo"
    if F > 1:
        tidx, _ = O.hamming_match(d2, kp, kc, W, 2 * f[:-1], 2 * f[:-1] + 2, -1, 0, 0, p.track_max_distance, H=H)
        if p.cross_check:
            tb, _ = O.hamming_match(d2, kp, kc, W, 2 * f[:-1] + 2, 2 * f[:-1], -1, 0, 0, p.track_max_distance, H=H)
            tidx = O.cross_check(tidx, tb)
        gt = res.track_idx.cpu().numpy()
        for r in range(F - 1):
            assert np.array_equal(gt[r, :kc[2 * r]], tidx[r, :kc[2 * r]]), tag + " track"
    print("ok  " + tag + f"  keypoints {kc.tolist()}", flush=True)
print(f"fuzz: {n_cases} cases bit-exact")
