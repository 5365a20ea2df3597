"""GPU parity: every front-end HIP kernel, called through the C ABI, against the CPU oracle on the
same inputs.  Integer work -> bit-exact (candidates compare as sets: the GPU appends in any order)."""
import numpy as np
import pytest
import torch

from visual_underwater_slam_amd import synth

pytestmark = pytest.mark.gpu


_KEEP = []   # device inputs stay referenced until the test module is done: a freed tensor's memory
             # would be handed to the next allocation while a launched kernel still reads it


def _dev(a):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    _KEEP.append(t)
    if len(_KEEP) > 64:
        torch.cuda.synchronize()
        del _KEEP[:32]
    return t


def _u32(t):
    return t.cpu().numpy().view(np.uint32)


def _call(name, *args):
    import visual_underwater_slam_amd._lib as L
    L.call(name, *args)


def gpu_fast_score(img, thr):
    import visual_underwater_slam_amd._lib as L
    d = _dev(img)
    n, H, W = img.shape
    out = torch.empty((n, H, W), dtype=torch.uint8, device="cuda")
    L.call("vus_fast_score", d.data_ptr(), n, H, W, W, thr, out.data_ptr(), L.current_stream_ptr())
    torch.cuda.synchronize()
    return out.cpu().numpy()


def gpu_detect(img, thr=10, border=31, cap=32768, want_blur=True):
    import visual_underwater_slam_amd._lib as L
    d = _dev(img)
    n, H, W = img.shape
    keys = torch.full((n, cap), -1, dtype=torch.int32, device="cuda")
    cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
    blur = torch.empty((n, H, W), dtype=torch.uint8, device="cuda") if want_blur else None
    L.call("vus_fast_detect", d.data_ptr(), n, H, W, W, thr, border, L.ptr(blur), keys.data_ptr(), cap,
           cnt.data_ptr(), L.current_stream_ptr())
    torch.cuda.synchronize()
    return keys, cnt, blur


@pytest.mark.parametrize("shape", [(7, 7), (33, 70), (96, 160), (100, 131), (720, 1280)])
def test_fast_score_and_blur_bit_exact(gpu, oracle, shape):
    H, W = shape
    if H <= 720 and W <= 1280 and H >= 16:
        img = synth.stereo_frames(2, 1, H=H, W=W)[0]
    else:
        img = np.random.default_rng(0).integers(0, 256, (2, H, W), dtype=np.uint8)
    for thr in (10, 40):
        assert np.array_equal(gpu_fast_score(img, thr), oracle.fast_score(img, thr))
    import visual_underwater_slam_amd._lib as L
    d = _dev(img)
    out = torch.empty((2, H, W), dtype=torch.uint8, device="cuda")
    L.call("vus_blur7", d.data_ptr(), 2, H, W, W, out.data_ptr(), L.current_stream_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), oracle.blur7(img))


def test_fast_score_random_noise_and_pitch(gpu, oracle):
    """Uniform noise is the densest-corner case; also exercises pitch > W."""
    import visual_underwater_slam_amd._lib as L
    rng = np.random.default_rng(5)
    H, W, pitch = 75, 90, 128
    buf = rng.integers(0, 256, (3, H, pitch), dtype=np.uint8)
    img = np.ascontiguousarray(buf[:, :, :W])
    d = _dev(buf)
    out = torch.empty((3, H, W), dtype=torch.uint8, device="cuda")
    L.call("vus_fast_score", d.data_ptr(), 3, H, W, pitch, 12, out.data_ptr(), L.current_stream_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), oracle.fast_score(img, 12))


@pytest.mark.parametrize("shape,nf", [((128, 192), 3), ((720, 1280), 2)])
def test_detect_candidate_sets_bit_exact(gpu, oracle, shape, nf):
    H, W = shape
    img = synth.stereo_frames(11, nf, H=H, W=W).reshape(2 * nf, H, W)
    keys, cnt, blur = gpu_detect(img)
    okeys, ocnt, oblur = oracle.fast_detect(img)
    assert np.array_equal(cnt.cpu().numpy(), ocnt)
    g = _u32(keys)
    for n in range(2 * nf):
        assert np.array_equal(np.sort(g[n, :ocnt[n]]), np.sort(okeys[n, :ocnt[n]]))
    assert np.array_equal(blur.cpu().numpy(), oblur)


def test_detect_flat_image_and_overflow(gpu, oracle):
    flat = np.full((2, 100, 120), 50, np.uint8)
    keys, cnt, _ = gpu_detect(flat, want_blur=False)
    assert cnt.cpu().tolist() == [0, 0]
    img = synth.stereo_frames(0, 1, H=256, W=256).reshape(2, 256, 256)
    _, ocnt, _ = oracle.fast_detect(img, cand_cap=32768)
    keys, cnt, _ = gpu_detect(img, cap=16, want_blur=False)      # far too small on purpose
    assert np.array_equal(cnt.cpu().numpy(), ocnt)               # true count is still reported
    assert (ocnt > 16).all()
    okeys, _, _ = oracle.fast_detect(img)
    g = _u32(keys)
    for n in range(2):                                           # whatever was kept is a real key
        assert set(g[n].tolist()) <= set(okeys[n, :ocnt[n]].tolist())


@pytest.mark.parametrize("max_kp", [1, 37, 500, 2000, 4096])
def test_select_topk_bit_exact(gpu, oracle, max_kp):
    import visual_underwater_slam_amd._lib as L
    img = synth.stereo_frames(4, 2, H=360, W=640).reshape(4, 360, 640)
    okeys, ocnt, _ = oracle.fast_detect(img, want_blur=False)
    ocnt2 = ocnt.copy()
    ocnt2[3] = 20             # fewer candidates than max_kp
    ocnt2[2] = 0              # none at all
    rng = np.random.default_rng(2)
    shuffled = okeys.copy()
    for n in range(4):
        shuffled[n, :ocnt2[n]] = rng.permutation(okeys[n, :ocnt2[n]])
    kp = torch.empty((4, max_kp), dtype=torch.int32, device="cuda")
    kc = torch.empty(4, dtype=torch.int32, device="cuda")
    L.call("vus_select_topk", _dev(shuffled.view(np.int32)).data_ptr(), _dev(ocnt2).data_ptr(), 4,
           okeys.shape[1], max_kp, kp.data_ptr(), kc.data_ptr(), L.current_stream_ptr())
    torch.cuda.synchronize()
    ekp, ekc = oracle.select_topk(okeys, ocnt2, max_kp)
    assert np.array_equal(kc.cpu().numpy(), ekc)
    assert np.array_equal(_u32(kp), ekp)


def test_select_topk_many_equal_scores(gpu, oracle):
    """All candidates share one score: the cut falls inside the raster-order tie-break."""
    import visual_underwater_slam_amd._lib as L
    rng = np.random.default_rng(9)
    pos = rng.permutation(500000)[:30000].astype(np.uint32)
    keys = ((np.uint32(255 - 77) << 24) | pos)[None]
    cnt = np.array([30000], np.int32)
    kp = torch.empty((1, 2000), dtype=torch.int32, device="cuda")
    kc = torch.empty(1, dtype=torch.int32, device="cuda")
    L.call("vus_select_topk", _dev(keys.view(np.int32)).data_ptr(), _dev(cnt).data_ptr(), 1, 30000, 2000,
           kp.data_ptr(), kc.data_ptr(), L.current_stream_ptr())
    torch.cuda.synchronize()
    ekp, _ = oracle.select_topk(keys, cnt, 2000)
    assert np.array_equal(_u32(kp), ekp)
    assert np.array_equal(ekp[0] & 0xFFFFFF, np.sort(pos)[:2000])


def _pipeline_oracle(oracle, img, max_kp):
    keys, cnt, blur = oracle.fast_detect(img)
    kp, kc = oracle.select_topk(keys, cnt, max_kp)
    desc, ang = oracle.orient_rbrief(img, blur, kp, kc)
    return kp, kc, blur, desc, ang


@pytest.mark.parametrize("shape,max_kp", [((160, 224), 300), ((720, 1280), 2000), ((100, 236), 150), ((97, 203), 120)])
def test_orient_rbrief_bit_exact(gpu, oracle, shape, max_kp):
    """(97, 203): rows not dword aligned (exact-start patch loads); max_kp values that leave the last wave's batch of
    eight keypoints partly empty."""
    import visual_underwater_slam_amd._lib as L
    H, W = shape
    img = synth.stereo_frames(20, 1, H=H, W=W).reshape(2, H, W)
    kp, kc, blur, edesc, eang = _pipeline_oracle(oracle, img, max_kp)
    kc2 = kc.copy()
    kc2[1] = min(kc2[1], 5)     # mostly-empty image: unused slots must come out zeroed
    edesc, eang = oracle.orient_rbrief(img, blur, kp, kc2)
    desc = torch.empty((2, max_kp, 4), dtype=torch.int64, device="cuda")
    ang = torch.empty((2, max_kp), dtype=torch.uint8, device="cuda")
    L.call("vus_orient_rbrief", _dev(img).data_ptr(), _dev(blur).data_ptr(), 2, H, W, W,
           _dev(kp.view(np.int32)).data_ptr(), _dev(kc2).data_ptr(), max_kp, desc.data_ptr(), ang.data_ptr(),
           L.current_stream_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(ang.cpu().numpy(), eang)
    assert np.array_equal(desc.cpu().numpy().view(np.uint64), edesc)


@pytest.mark.parametrize("shape,max_kp", [((160, 224), 300), ((720, 1280), 2000), ((97, 203), 120)])
def test_orient_rbrief_in_cell_order_is_the_same_result(gpu, oracle, shape, max_kp):
    """vus_orient_order groups an image's keypoints by 64 x 64 cell (a permutation that fixes the unused slots);
    vus_orient_rbrief_ordered serves the slots in that order -- and in ANY such order -- with bit-identical outputs."""
    import visual_underwater_slam_amd._lib as L
    H, W = shape
    img = synth.stereo_frames(21, 1, H=H, W=W).reshape(2, H, W)
    kp, kc, blur, _, _ = _pipeline_oracle(oracle, img, max_kp)
    kc2 = kc.copy()
    kc2[1] = min(kc2[1], 37)     # a partly filled last batch of eight, then unused slots
    edesc, eang = oracle.orient_rbrief(img, blur, kp, kc2)
    d_kp, d_kc = _dev(kp.view(np.int32)), _dev(kc2)
    order = torch.full((2, max_kp), -7, dtype=torch.int32, device="cuda")
    L.call("vus_orient_order", d_kp.data_ptr(), d_kc.data_ptr(), 2, max_kp, H, W, order.data_ptr(), L.current_stream_ptr())
    torch.cuda.synchronize()
    o = order.cpu().numpy()
    cw = (W + 63) // 64
    for n in range(2):
        c = int(kc2[n])
        assert sorted(o[n, :c].tolist()) == list(range(c)) and np.array_equal(o[n, c:], np.arange(c, max_kp))
        pos = (kp[n, o[n, :c]] & 0xFFFFFF).astype(np.int64)
        cell = (pos // W // 64) * cw + (pos % W) // 64
        assert np.all(np.diff(cell) >= 0), "keypoints are not grouped by cell in raster order"
    rng = np.random.default_rng(3)
    shuffled = o.copy()
    for n in range(2):
        c = int(kc2[n])
        shuffled[n, :c] = rng.permutation(c)
    for perm in (o, shuffled):
        desc = torch.full((2, max_kp, 4), -1, dtype=torch.int64, device="cuda")
        ang = torch.full((2, max_kp), 99, dtype=torch.uint8, device="cuda")
        L.call("vus_orient_rbrief_ordered", _dev(img).data_ptr(), _dev(blur).data_ptr(), 2, H, W, W, d_kp.data_ptr(), d_kc.data_ptr(),
               max_kp, _dev(perm).data_ptr(), desc.data_ptr(), ang.data_ptr(), L.current_stream_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(ang.cpu().numpy(), eang)
        assert np.array_equal(desc.cpu().numpy().view(np.uint64), edesc)


def test_orient_rbrief_crowded_keypoints_and_duplicates(gpu, oracle):
    """Hundreds of keypoints inside one 128 x 48 patch of the image (duplicates among them) and a second image with one
    keypoint in each corner of such patches: keypoints of a wave's batch of eight share rows and lines."""
    import visual_underwater_slam_amd._lib as L
    H, W, K = 200, 400, 333
    rng = np.random.default_rng(12)
    img = rng.integers(0, 256, (2, H, W), dtype=np.uint8)
    blur = oracle.blur7(img)
    ys, xs = rng.integers(48, 96, K), rng.integers(128, 256, K)
    kp = np.zeros((2, K), np.uint32)
    kp[0] = (np.uint32(77) << 24) | (ys * W + xs).astype(np.uint32)
    corners = [(y, x) for y in (0, 47, 48, 95, 96, 199) for x in (0, 127, 128, 255, 256, 399)]
    kp[1, :len(corners)] = [(50 << 24) | (y * W + x) for y, x in corners]
    kc = np.array([K, len(corners)], np.int32)
    edesc, eang = oracle.orient_rbrief(img, blur, kp, kc)
    desc = torch.empty((2, K, 4), dtype=torch.int64, device="cuda")
    ang = torch.empty((2, K), dtype=torch.uint8, device="cuda")
    L.call("vus_orient_rbrief", _dev(img).data_ptr(), _dev(blur).data_ptr(), 2, H, W, W,
           _dev(kp.view(np.int32)).data_ptr(), _dev(kc).data_ptr(), K, desc.data_ptr(), ang.data_ptr(),
           L.current_stream_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(ang.cpu().numpy(), eang)
    assert np.array_equal(desc.cpu().numpy().view(np.uint64), edesc)


def test_orient_rbrief_keypoints_at_image_edge_are_clamped(gpu, oracle):
    """Keys right at the border (outside the detector's own border filter) stay memory-safe and equal."""
    import visual_underwater_slam_amd._lib as L
    H, W = 64, 80
    img = np.random.default_rng(4).integers(0, 256, (1, H, W), dtype=np.uint8)
    blur = oracle.blur7(img)
    pts = [(0, 0), (0, W - 1), (H - 1, 0), (H - 1, W - 1), (5, 40), (H - 2, 17)]
    kp = np.array([[(100 << 24) | (y * W + x) for y, x in pts]], np.uint32)
    kc = np.array([len(pts)], np.int32)
    edesc, eang = oracle.orient_rbrief(img, blur, kp, kc)
    desc = torch.empty((1, len(pts), 4), dtype=torch.int64, device="cuda")
    ang = torch.empty((1, len(pts)), dtype=torch.uint8, device="cuda")
    L.call("vus_orient_rbrief", _dev(img).data_ptr(), _dev(blur).data_ptr(), 1, H, W, W,
           _dev(kp.view(np.int32)).data_ptr(), _dev(kc).data_ptr(), len(pts), desc.data_ptr(), ang.data_ptr(),
           L.current_stream_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(ang.cpu().numpy(), eang)
    assert np.array_equal(desc.cpu().numpy().view(np.uint64), edesc)


@pytest.mark.parametrize("gate", [(-1, 0, 0, 256), (5, 0, 128, 64), (2, -3, 40, 30)])
def test_hamming_match_bit_exact(gpu, oracle, gate):
    import visual_underwater_slam_amd._lib as L
    H, W, K = 360, 640, 1500     # 1500 > one 1024-descriptor LDS tile
    img = synth.stereo_frames(30, 2, H=H, W=W).reshape(4, H, W)
    kp, kc, blur, desc, ang = _pipeline_oracle(oracle, img, K)
    kc = kc.copy()
    kc[3] = 700
    q = np.array([0, 0, 2, 3, 1], np.int32)
    t = np.array([1, 2, 3, 2, 1], np.int32)
    max_dy, mind, maxd, maxdist = gate
    eidx, edist = oracle.hamming_match(desc, kp, kc, W, q, t, max_dy, mind, maxd, maxdist)
    idx = torch.empty((len(q), K), dtype=torch.int32, device="cuda")
    dist = torch.empty((len(q), K), dtype=torch.int32, device="cuda")
    L.call("vus_hamming_match", _dev(desc.view(np.int64)).data_ptr(), _dev(kp.view(np.int32)).data_ptr(),
           _dev(kc).data_ptr(), K, H, W, _dev(q).data_ptr(), _dev(t).data_ptr(), len(q), max_dy, mind, maxd,
           maxdist, idx.data_ptr(), dist.data_ptr(), L.current_stream_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(idx.cpu().numpy(), eidx)
    assert np.array_equal(dist.cpu().numpy(), edist)
    if max_dy < 0:
        assert (eidx[4, :kc[1]] == np.arange(kc[1])).all()     # self-match: identity, distance 0
        assert (edist[4, :kc[1]] == 0).all()


@pytest.mark.parametrize("max_dist", [256, 40])
def test_ungated_matcher_edge_cases_random_descriptors(gpu, oracle, max_dist):
    """The matrix-core matcher on synthetic descriptor sets: duplicates (ties -> lowest index), all-zero and
    all-one descriptors, counts that are 0 / 1 / not multiples of the 32-row tiles, fewer trains than queries."""
    import visual_underwater_slam_amd._lib as L
    rng = np.random.default_rng(11)
    K, n_img = 300, 6
    desc = rng.integers(0, 2**63, size=(n_img, K, 4), dtype=np.int64).view(np.uint64)
    desc ^= rng.integers(0, 2, size=desc.shape, dtype=np.uint64) << np.uint64(63)
    desc[1, 5] = desc[1, 2]                      # duplicates inside a train set: the lower index must win
    desc[1, 200] = desc[1, 2]
    desc[0, 7] = desc[1, 2]
    desc[0, 8] = 0
    desc[0, 9] = np.uint64(0xFFFFFFFFFFFFFFFF)
    desc[1, 33] = 0
    desc[1, 64] = np.uint64(0xFFFFFFFFFFFFFFFF)
    desc[2, :, :] = desc[0, :, :]                # image 2 == image 0 except for bit flips of growing weight
    for i in range(K):
        w = rng.integers(0, 256, size=i % 60)
        for b in w:
            desc[2, i, b // 64] ^= np.uint64(1) << np.uint64(b % 64)
    kc = np.array([300, 257, 300, 1, 0, 33], np.int32)
    kp = np.zeros((n_img, K), np.uint32)          # positions are not used by the ungated matcher
    q = np.array([0, 1, 0, 0, 0, 5, 3, 4, 2], np.int32)
    t = np.array([1, 0, 2, 3, 4, 1, 0, 0, 0], np.int32)
    eidx, edist = oracle.hamming_match(desc, kp, kc, 64, q, t, -1, 0, 0, max_dist, H=64)
    idx = torch.empty((len(q), K), dtype=torch.int32, device="cuda")
    dist = torch.empty((len(q), K), dtype=torch.int32, device="cuda")
    L.call("vus_hamming_match", _dev(desc.view(np.int64)).data_ptr(), _dev(kp.view(np.int32)).data_ptr(),
           _dev(kc).data_ptr(), K, 64, 64, _dev(q).data_ptr(), _dev(t).data_ptr(), len(q), -1, 0, 0, max_dist,
           idx.data_ptr(), dist.data_ptr(), L.current_stream_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(idx.cpu().numpy(), eidx)
    assert np.array_equal(dist.cpu().numpy(), edist)
    assert eidx[0, 7] == 2 and edist[0, 7] == 0                      # three identical trains: index 2 wins
    assert (eidx[4] == -1).all() and (edist[4] == 512).all()         # empty train set
    assert (eidx[7] == -1).all()                                     # empty query set
    if max_dist == 256:
        assert (eidx[3, :300] == 0).all()                            # single train descriptor


def test_triangulate_matches_oracle(gpu, oracle):
    from visual_underwater_slam_amd import frontend
    rng = np.random.default_rng(8)
    n = 1000
    u0 = rng.uniform(-0.9, 0.9, n); v0 = rng.uniform(-0.9, 0.9, n)
    feat = np.stack([u0, v0, u0 + rng.uniform(0.004, 0.06, n), v0 + rng.uniform(-1e-3, 1e-3, n)], 1)
    cam = np.array([1827.0, 1827.5999755859375, 968.9000244140625, 561.4000244140625, 0.063, 1920, 1080, 0])
    A = rng.normal(size=(3, 3)); Q, _ = np.linalg.qr(A)
    Rt = np.concatenate([Q.reshape(-1), rng.normal(size=3)])
    got = frontend.triangulate(_dev(feat), _dev(cam), _dev(Rt)).cpu().numpy()
    exp = oracle.triangulate(feat, cam, Rt)
    assert np.array_equal(got, exp)      # same operation order, no FMA contraction on either side


def test_full_frontend_pipeline_matches_oracle_chain(gpu, oracle):
    """StereoOrbFrontend.process() end to end on 3 full-size frames == the oracle stage by stage."""
    from visual_underwater_slam_amd.frontend import StereoOrbFrontend, ImageProcessorParams
    F, H, W = 3, 720, 1280
    img = synth.stereo_frames(100, F)
    fe = StereoOrbFrontend(H, W, max_frames=4, params=ImageProcessorParams())
    res = fe.process(torch.from_numpy(img).cuda())
    torch.cuda.synchronize()
    p = fe.p
    flat = img.reshape(2 * F, H, W)
    kp, kc, blur, desc, ang = _pipeline_oracle(oracle, flat, p.max_features)
    assert np.array_equal(res.kp_count.cpu().numpy(), kc)
    assert np.array_equal(_u32(res.kp_keys), kp)                 # bit-exact FAST keypoint indices
    assert np.array_equal(res.desc.cpu().numpy().view(np.uint64), desc)
    f = np.arange(F, dtype=np.int32)
    sidx, sdist = oracle.hamming_match(desc, kp, kc, W, 2 * f, 2 * f + 1, p.stereo_threshold,
                                       p.min_disparity, p.max_disparity, p.stereo_max_distance)
    tidx, tdist = oracle.hamming_match(desc, kp, kc, W, 2 * f[:-1], 2 * f[:-1] + 2, -1, 0, 0,
                                       p.track_max_distance)
    assert np.array_equal(res.stereo_idx.cpu().numpy(), sidx)    # bit-exact match pairs
    assert np.array_equal(res.stereo_dist.cpu().numpy(), sdist)
    assert np.array_equal(res.track_idx.cpu().numpy(), tidx)
    assert np.array_equal(res.track_dist.cpu().numpy(), tdist)
    # the emitted CameraMeasurement records carry the reference's field names and NDC convention
    msgs = fe.camera_measurements(res)
    assert len(msgs) == F and len(msgs[0].features) == int((sidx[0] >= 0).sum())
    ft = msgs[1].features[0]
    assert -1.0 <= ft.u0 <= 1.0 and -1.0 <= ft.v1 <= 1.0 and ft.id >= 0
    ids0 = {f_.id for f_ in msgs[0].features}
    assert len(ids0 & {f_.id for f_ in msgs[1].features}) > 50   # tracks persist across frames


def test_configs1_launch_shape_1000_frames_matches_oracle_at_both_ends_and_the_middle(gpu, oracle):
    """BASELINE.json configs[1] exactly as bench.py runs it -- ONE StereoOrbFrontend.process() over the resident
    1000-frame 1280x720 stream (grid.z = 2000 images, 1.84 GB of pixels, 2000 x 32768 candidate slots) -- compared with
    the oracle on frames 0, 499 and 999 (and their temporal successors): keypoints, descriptors, stereo and temporal
    matches bit for bit.  The timed launch shape, not a 3-frame stand-in."""
    from visual_underwater_slam_amd.frontend import StereoOrbFrontend, ImageProcessorParams
    F, H, W = 1000, 720, 1280
    cv = synth.canvas(torch, "cuda")
    stream = torch.empty((F, 2, H, W), dtype=torch.uint8, device="cuda")
    for s0 in range(0, F, 8):
        stream[s0:s0 + 8] = synth.stereo_frames(s0, min(8, F - s0), H, W, xp=torch, device="cuda", canvas_arr=cv)
    fe = StereoOrbFrontend(H, W, max_frames=F, params=ImageProcessorParams())
    res = fe.process(stream)
    torch.cuda.synchronize()
    p = fe.p
    assert int(res.kp_count.min()) == p.max_features                  # every image yields its 2000 keypoints
    for t in (0, 499, 998):                                            # frames t, t + 1: 0/1, 499/500, 998/999
        img = stream[t:t + 2].cpu().numpy()
        assert np.array_equal(img, synth.stereo_frames(t, 2))          # the torch-generated stream == the numpy one
        flat = img.reshape(4, H, W)
        kp, kc, blur, desc, ang = _pipeline_oracle(oracle, flat, p.max_features)
        sl = slice(2 * t, 2 * t + 4)
        assert np.array_equal(res.kp_count[sl].cpu().numpy(), kc)
        assert np.array_equal(_u32(res.kp_keys[sl]), kp)
        assert np.array_equal(res.desc[sl].cpu().numpy().view(np.uint64), desc)
        assert np.array_equal(res.angle[sl].cpu().numpy(), ang)
        f = np.arange(2, dtype=np.int32)
        sidx, sdist = oracle.hamming_match(desc, kp, kc, W, 2 * f, 2 * f + 1, p.stereo_threshold,
                                           p.min_disparity, p.max_disparity, p.stereo_max_distance)
        tidx, tdist = oracle.hamming_match(desc, kp, kc, W, 2 * f[:1], 2 * f[:1] + 2, -1, 0, 0, p.track_max_distance)
        assert np.array_equal(res.stereo_idx[t:t + 2].cpu().numpy(), sidx)
        assert np.array_equal(res.stereo_dist[t:t + 2].cpu().numpy(), sdist)
        assert np.array_equal(res.track_idx[t:t + 1].cpu().numpy(), tidx)
        assert np.array_equal(res.track_dist[t:t + 1].cpu().numpy(), tdist)
        assert (sidx >= 0).sum() > 1500 and (tidx >= 0).sum() > 800


def _adaptive_images(H, W):
    """Images that stress the adaptive detector's estimate: textured frames, noise, almost no corners, none at all, and
    texture confined to one corner of the image (the tile sample sees little of it)."""
    rng = np.random.default_rng(11)
    tex = synth.stereo_frames(40, 1, H=H, W=W)[0]                                  # two corner-rich images
    noise = rng.integers(0, 256, (H, W), dtype=np.uint8)
    ramp = np.add.outer(np.arange(H), np.arange(W)).astype(np.float64)
    smooth = (ramp / ramp.max() * 200).astype(np.uint8)
    few = smooth.copy()
    for k in range(40):                                                            # 40 bright squares: ~160 corners
        y, x = int(rng.integers(40, H - 50)), int(rng.integers(40, W - 50))
        few[y:y + 9, x:x + 9] = 255
    blank = np.full((H, W), 77, np.uint8)
    corner = smooth.copy()
    corner[H - 200:, W - 420:] = tex[0][:200, :420]                                 # all the texture in one corner
    return np.stack([tex[0], tex[1], noise, few, blank, corner])


@pytest.mark.parametrize("shape", [(720, 1280), (360, 642)])
def test_adaptive_detector_selects_the_same_keypoints_as_fast_threshold(gpu, oracle, shape):
    """StereoOrbFrontend with the adaptive detector (per-image threshold from a tile sample, verified on the device,
    failed images detected again at fast_threshold) == the plain detector: keypoints, descriptors and matches bit for bit,
    whatever the estimate was worth."""
    from visual_underwater_slam_amd.frontend import StereoOrbFrontend, ImageProcessorParams
    H, W = shape
    K = 2000 if H == 720 else 500
    imgs = _adaptive_images(H, W)
    frames = torch.from_numpy(imgs.reshape(3, 2, H, W)).cuda()
    out = {}
    for adaptive in (True, False):
        fe = StereoOrbFrontend(H, W, max_frames=3, params=ImageProcessorParams(max_features=K, adaptive_fast=adaptive,
                                                                               cand_cap=131072))     # noise: 84 k candidates
        res = fe.process(frames)
        torch.cuda.synchronize()
        out[adaptive] = (fe, {k: getattr(res, k).cpu().numpy().copy() for k in
                              ("kp_keys", "kp_count", "desc", "angle", "stereo_idx", "stereo_dist", "track_idx", "track_dist")})
    for k, v in out[True][1].items():
        assert np.array_equal(v, out[False][1][k]), k
    fe = out[True][0]
    thr, retried = fe.fast_thr.cpu().numpy(), fe.fast_retry_list.cpu().numpy()[:int(fe.fast_retry_count.item())]
    assert (thr[:3] > 40).all()                     # corner-rich images: pruning is really on
    assert thr[4] == 10 and thr[3] == 10            # too few corners for a threshold above fast_threshold
    cnt = out[True][1]["kp_count"]
    assert cnt[4] == 0 and 0 < cnt[3] < K
    # the oracle's twin of the estimate: same histogram, same thresholds
    ohist, othr = oracle.fast_threshold_estimate(imgs, 10, 31, K, fe.p.fast_sample_stride)
    assert np.array_equal(fe.fast_hist.cpu().numpy(), ohist) and np.array_equal(thr, othr)
    # what the check must have sent back: images whose adaptive pass cannot yield K candidates
    okeys, ocnt, oretry = oracle.fast_detect_adaptive(imgs, othr, 10, 31, K, cand_cap=fe.p.cand_cap)
    assert sorted(retried.tolist()) == sorted(oretry.tolist())
    kp, kc = oracle.select_topk(okeys, ocnt, K)
    assert np.array_equal(_u32(fe.kp_keys[:6]), kp) and np.array_equal(cnt, kc)


def test_adaptive_detection_sublists_compact_and_report_overflow(gpu, oracle):
    """vus_fast_detect_adaptive fills an image's candidate list as eight sub-lists and compacts them: with room, the
    list is the oracle's candidate SET and cand_count its size; when a sub-list outgrows cand_cap / 8 - 1 the call
    reports an overflow (cand_count > cand_cap), every key it kept is a true candidate and the tail is VUS_KEY_INVALID."""
    import visual_underwater_slam_amd._lib as L
    H, W = 360, 640
    imgs = _adaptive_images(H, W)[:3]
    n = len(imgs)
    d_img = torch.from_numpy(imgs).cuda()
    thr_img = torch.tensor([12, 10, 30][:n], dtype=torch.int32, device="cuda")
    blur = torch.empty((n, H, W), dtype=torch.uint8, device="cuda")
    st = L.current_stream_ptr()
    want = []
    for i in range(n):
        ok, oc, _ = oracle.fast_detect(imgs[i:i + 1], int(thr_img[i]), 31, H * W, want_blur=False)
        want.append(set(ok[0, :int(oc[0])].tolist()))
    for cap in (H * W, 512):
        keys = torch.full((n, cap), 123, dtype=torch.int32, device="cuda")
        cnt = torch.zeros((n,), dtype=torch.int32, device="cuda")
        L.call("vus_fast_detect_adaptive", d_img.data_ptr(), n, H, W, W, thr_img.data_ptr(), 31, blur.data_ptr(), keys.data_ptr(), cap,
               cnt.data_ptr(), st)
        torch.cuda.synchronize()
        k, c = _u32(keys), cnt.cpu().numpy()
        for i in range(n):
            if cap == H * W:
                assert c[i] == len(want[i]) and set(k[i, :c[i]].tolist()) == want[i]
            elif len(want[i]) > cap // 8 - 1:      # more than one sub-list can hold: at least one of them overflowed here
                assert c[i] > cap, "overflow of a sub-list must be reported as an overflow of the list"
                valid = k[i][k[i] != 0xFFFFFFFF]
                assert len(valid) == len(set(valid.tolist())) and set(valid.tolist()) <= want[i] and len(valid) >= cap // 8 - 1
                assert np.all(k[i, len(valid):] == 0xFFFFFFFF) and np.all(k[i, :len(valid)] != 0xFFFFFFFF)


def test_adaptive_detector_check_catches_a_wrong_estimate(gpu, oracle):
    """vus_fast_detect_retry IS the guarantee: with thresholds far too high for every image (250), all of them fail the
    count check, are listed and detected again at fast_threshold -- the candidates of vus_fast_detect, exactly."""
    import visual_underwater_slam_amd._lib as L
    H, W, K, cap = 360, 640, 800, 32768
    imgs = _adaptive_images(H, W)[:4]
    n = len(imgs)
    d_img = torch.from_numpy(imgs).cuda()
    thr_img = torch.full((n,), 250, dtype=torch.int32, device="cuda")
    keys = torch.zeros((n, cap), dtype=torch.int32, device="cuda")
    cnt = torch.zeros((n,), dtype=torch.int32, device="cuda")
    lst = torch.zeros((n,), dtype=torch.int32, device="cuda"); m = torch.zeros((1,), dtype=torch.int32, device="cuda")
    st = L.current_stream_ptr()
    L.call("vus_fast_detect_adaptive", d_img.data_ptr(), n, H, W, W, thr_img.data_ptr(), 31, None, keys.data_ptr(), cap, cnt.data_ptr(), st)
    first = cnt.cpu().numpy().copy()
    L.call("vus_fast_detect_retry", d_img.data_ptr(), n, H, W, W, 10, thr_img.data_ptr(), K, 31, keys.data_ptr(), cap, cnt.data_ptr(),
           lst.data_ptr(), m.data_ptr(), st)
    torch.cuda.synchronize()
    assert (first < K).all() and int(m.item()) == n and sorted(lst.cpu().tolist()) == list(range(n))
    ekeys, ecnt, _ = oracle.fast_detect(imgs, thr=10, border=31, cand_cap=cap, want_blur=False)
    assert np.array_equal(cnt.cpu().numpy(), ecnt)
    for i in range(n):
        assert np.array_equal(np.sort(_u32(keys[i])[:ecnt[i]]), np.sort(ekeys[i][:ecnt[i]]))


def test_track_ids_matches_oracle_and_feeds_get_landmarks(gpu, oracle):
    """vus_track_ids on the GPU == oracle; its features go through vus_triangulate (batch.py:144-176)."""
    from visual_underwater_slam_amd.frontend import StereoOrbFrontend, ImageProcessorParams, triangulate
    F, H, W = 6, 360, 640
    img = synth.stereo_frames(200, F, H=H, W=W)
    fe = StereoOrbFrontend(H, W, max_frames=F, params=ImageProcessorParams(max_features=800))
    res = fe.process(torch.from_numpy(img).cuda())
    ids, feats, n_ids = fe.feature_tracks(res)
    torch.cuda.synchronize()
    eids, efeat, en = oracle.track_ids(res.stereo_idx.cpu().numpy(), res.track_idx.cpu().numpy(),
                                       res.kp_keys.cpu().numpy().view(np.uint32), res.kp_count.cpu().numpy(), H, W)
    assert n_ids == en and np.array_equal(ids.cpu().numpy(), eids)
    assert np.array_equal(feats.cpu().numpy(), efeat)
    pub = eids >= 0
    assert pub.sum() > 500 and len(set(eids[0][pub[0]].tolist())) == pub[0].sum()      # unique per frame
    persisted = len(set(eids[0][pub[0]].tolist()) & set(eids[1][pub[1]].tolist()))
    assert persisted > 50
    msgs = fe.camera_measurements(res)
    assert [len(m.features) for m in msgs] == pub.sum(1).tolist()
    # feed the published features of frame 0 through get_landmarks on the GPU
    cam = torch.tensor([1827.0, 1827.5999755859375, 968.9000244140625, 561.4000244140625, 0.063, 1920, 1080, 0],
                       dtype=torch.float64, device="cuda")
    Rt = torch.tensor([1.0, 0, 0, 0, 1, 0, 0, 0, 1, 0.5, -0.25, 2.0], dtype=torch.float64, device="cuda")
    f0 = feats[0][ids[0] >= 0].contiguous()
    lm = triangulate(f0, cam, Rt).cpu().numpy()
    exp = oracle.triangulate(f0.cpu().numpy(), cam.cpu().numpy(), Rt.cpu().numpy())
    assert np.array_equal(lm, exp)


def test_pipeline_edge_cases_single_frame_and_featureless_images(gpu, oracle):
    """F = 1 (no temporal pair), one camera seeing nothing, and an image size that is no multiple of the tile."""
    from visual_underwater_slam_amd.frontend import StereoOrbFrontend, ImageProcessorParams
    H, W = 250, 333
    tex = synth.stereo_frames(9, 1, H=H, W=320)[0, 0]
    left = np.zeros((H, W), np.uint8); left[:, :320] = tex; left[:, 320:] = tex[:, :13]
    img = np.stack([left, np.full((H, W), 90, np.uint8)])[None]          # right camera: flat -> 0 keypoints
    fe = StereoOrbFrontend(H, W, max_frames=2, params=ImageProcessorParams(max_features=300))
    res = fe.process(torch.from_numpy(img).cuda())
    ids, feats, n_ids = fe.feature_tracks(res)
    torch.cuda.synchronize()
    flat = img.reshape(2, H, W)
    kp, kc, blur, desc, ang = _pipeline_oracle(oracle, flat, 300)
    assert kc[1] == 0 and kc[0] > 50
    assert np.array_equal(res.kp_count.cpu().numpy(), kc) and np.array_equal(_u32(res.kp_keys), kp)
    assert np.array_equal(res.desc.cpu().numpy().view(np.uint64), desc)
    assert (res.stereo_idx.cpu().numpy() == -1).all() and (res.stereo_dist.cpu().numpy() == 512).all()
    assert res.track_idx.shape[0] == 0 and n_ids == 0 and (ids.cpu().numpy() == -1).all()
    assert fe.camera_measurements(res)[0].features == []


# ---------------------------------------------------------------------------------------------
# optional ORB scale pyramid
@pytest.mark.parametrize("src,dst", [((40, 56), (40, 56)), ((40, 56), (20, 28)), ((720, 1280), (600, 1067)),
                                      ((201, 357), (168, 298)), ((37, 53), (61, 90))])
def test_resize_bilinear_bit_exact(gpu, oracle, src, dst):
    import visual_underwater_slam_amd._lib as L
    rng = np.random.default_rng(src[0] + dst[1])
    img = rng.integers(0, 256, size=(3,) + src, dtype=np.uint8)
    d = _dev(img)
    pitch_d = dst[1] + 5                                     # destination rows are padded
    out = torch.zeros((3, dst[0], pitch_d), dtype=torch.uint8, device="cuda")
    L.call("vus_resize_bilinear", d.data_ptr(), 3, src[0], src[1], src[1], out.data_ptr(), dst[0], dst[1], pitch_d,
           L.current_stream_ptr())
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.array_equal(got[:, :, :dst[1]], oracle.resize_bilinear(img, *dst))
    assert not got[:, :, dst[1]:].any()                      # the padding is not written


def _pyramid_oracle(oracle, flat, p, H, W):
    """The oracle's stages chained level by level (host logic restated independently of frontend.py)."""
    from visual_underwater_slam_amd.frontend import pyramid_layout
    sizes, quotas = pyramid_layout(H, W, p.max_features, p.n_levels, p.scale_factor)
    m = oracle.new_merged(flat.shape[0], p.max_features)
    lvl = flat
    for l, ((h, w), q) in enumerate(zip(sizes, quotas)):
        if l > 0:
            lvl = oracle.resize_bilinear(lvl, h, w)
        ck, cc, blur = oracle.fast_detect(lvl, p.fast_threshold, p.border, p.cand_cap)
        kp, kc = oracle.select_topk(ck, cc, max(q, 1))
        desc, ang = oracle.orient_rbrief(lvl, blur, kp, kc)
        oracle.pyramid_append(kp, kc, desc, ang, h, w, l, H, W, m)
    return m


@pytest.mark.parametrize("shape,levels,kmax", [((240, 320), 4, 600), ((720, 1280), 8, 2000)])
def test_pyramid_pipeline_matches_oracle_chain(gpu, oracle, shape, levels, kmax):
    """n_levels > 1: merged keypoints, levels, sub-pixel positions, descriptors and both match sets are bit-exact."""
    from visual_underwater_slam_amd.frontend import StereoOrbFrontend, ImageProcessorParams
    H, W = shape
    F = 2
    img = synth.stereo_frames(7, F)[:, :, :H, :W].copy()
    p = ImageProcessorParams(n_levels=levels, max_features=kmax)
    fe = StereoOrbFrontend(H, W, max_frames=F, params=p)
    res = fe.process(torch.from_numpy(img).cuda())
    torch.cuda.synchronize()
    m = _pyramid_oracle(oracle, img.reshape(2 * F, H, W), fe.p, H, W)
    kc = m["kp_count"]
    assert np.array_equal(res.kp_count.cpu().numpy(), kc) and kc.min() > 100
    assert np.array_equal(_u32(res.kp_keys), m["kp_keys"])
    gd, ga = res.desc.cpu().numpy().view(np.uint64), res.angle.cpu().numpy()
    gl, gq = res.kp_level.cpu().numpy(), res.kp_xy_q4.cpu().numpy()
    for n in range(2 * F):
        c = kc[n]
        assert np.array_equal(gd[n, :c], m["desc"][n, :c]) and np.array_equal(ga[n, :c], m["angle"][n, :c])
        assert np.array_equal(gl[n, :c], m["kp_level"][n, :c]) and np.array_equal(gq[n, :c], m["kp_xy_q4"][n, :c])
    assert len(np.unique(m["kp_level"][0, :kc[0]])) == levels         # every level contributes
    f = np.arange(F, dtype=np.int32)
    desc = m["desc"].copy()
    sidx, sdist = oracle.hamming_match(desc, m["kp_keys"], kc, W, 2 * f, 2 * f + 1, p.stereo_threshold,
                                       p.min_disparity, p.max_disparity, p.stereo_max_distance, H=H)
    tidx, tdist = oracle.hamming_match(desc, m["kp_keys"], kc, W, 2 * f[:-1], 2 * f[:-1] + 2, -1, 0, 0,
                                       p.track_max_distance, H=H)

    def same_matches(gi, gdist, oi, od, q_imgs):
        for r, qi in enumerate(q_imgs):
            c = kc[qi]
            assert np.array_equal(gi[r, :c], oi[r, :c]) and np.array_equal(gdist[r, :c], od[r, :c])

    same_matches(res.stereo_idx.cpu().numpy(), res.stereo_dist.cpu().numpy(), sidx, sdist, 2 * f)
    same_matches(res.track_idx.cpu().numpy(), res.track_dist.cpu().numpy(), tidx, tdist, 2 * f[:-1])
    assert (sidx[0, :kc[0]] >= 0).sum() > 50
    msgs = fe.camera_measurements(res)
    assert len(msgs) == F and len(msgs[0].features) > 50


def test_cross_check_keeps_only_mutual_matches(gpu, oracle):
    """ImageProcessorParams(cross_check=True): forward matches filtered by the backward pairing == the oracle chain."""
    from visual_underwater_slam_amd.frontend import StereoOrbFrontend, ImageProcessorParams
    F, H, W, K = 3, 240, 320, 400
    img = synth.stereo_frames(21, F, H=H, W=W)
    p = ImageProcessorParams(max_features=K, cross_check=True)
    fe = StereoOrbFrontend(H, W, max_frames=F, params=p)
    res = fe.process(torch.from_numpy(img).cuda())
    torch.cuda.synchronize()
    kp, kc, blur, desc, ang = _pipeline_oracle(oracle, img.reshape(2 * F, H, W), K)
    f = np.arange(F, dtype=np.int32)
    fwd, _ = oracle.hamming_match(desc, kp, kc, W, 2 * f, 2 * f + 1, p.stereo_threshold, p.min_disparity, p.max_disparity,
                                  p.stereo_max_distance, H=H)
    bwd, _ = oracle.hamming_match(desc, kp, kc, W, 2 * f + 1, 2 * f, p.stereo_threshold, -p.max_disparity, -p.min_disparity,
                                  p.stereo_max_distance, H=H)
    want = oracle.cross_check(fwd, bwd)
    assert np.array_equal(res.stereo_idx.cpu().numpy(), want)
    tf, _ = oracle.hamming_match(desc, kp, kc, W, 2 * f[:-1], 2 * f[:-1] + 2, -1, 0, 0, p.track_max_distance, H=H)
    tb, _ = oracle.hamming_match(desc, kp, kc, W, 2 * f[:-1] + 2, 2 * f[:-1], -1, 0, 0, p.track_max_distance, H=H)
    assert np.array_equal(res.track_idx.cpu().numpy(), oracle.cross_check(tf, tb))
    kept, before = int((want >= 0).sum()), int((fwd >= 0).sum())
    assert 0 < kept <= before
    # mutual by construction
    for r in range(F):
        for i in np.nonzero(want[r] >= 0)[0][:50]:
            assert bwd[r, want[r, i]] == i


@pytest.mark.parametrize("grid", [(3, 4, 4, 64), (5, 7, 3, 50), (1, 1, 6, 6)])
def test_select_grid_bit_exact_and_in_the_pipeline(gpu, oracle, grid):
    """Grid-bucketed selection (the nodelet's grid_row / grid_col / grid_max_feature_num) == the oracle, both as
    a kernel and through StereoOrbFrontend."""
    import visual_underwater_slam_amd._lib as L
    from visual_underwater_slam_amd.frontend import StereoOrbFrontend, ImageProcessorParams
    gr, gc, per, K = grid
    F, H, W = 2, 240, 320
    img = synth.stereo_frames(33, F, H=H, W=W)
    flat = img.reshape(2 * F, H, W)
    ck, cc, blur = oracle.fast_detect(flat, 10, 31, 32768)
    want, wc = oracle.select_grid(ck, cc, H, W, gr, gc, per, K)
    keys = torch.empty((2 * F, K), dtype=torch.int32, device="cuda")
    cnt = torch.empty(2 * F, dtype=torch.int32, device="cuda")
    L.call("vus_select_grid", _dev(ck.view(np.int32)).data_ptr(), _dev(cc).data_ptr(), 2 * F, ck.shape[1], H, W, gr, gc,
           per, K, keys.data_ptr(), cnt.data_ptr(), L.current_stream_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(cnt.cpu().numpy(), wc) and np.array_equal(_u32(keys), want)
    assert wc.max() <= min(K, gr * gc * per) and wc.min() > 0
    # every kept keypoint lies in the cell its slot order says, at most `per` per cell
    pos = want[0, :wc[0]] & 0xFFFFFF
    cells = (pos // W) * gr // H * gc + (pos % W) * gc // W
    assert (np.diff(cells.astype(np.int64)) >= 0).all() and np.bincount(cells, minlength=gr * gc).max() <= per
    fe = StereoOrbFrontend(H, W, max_frames=F, params=ImageProcessorParams(max_features=K, grid_row=gr, grid_col=gc,
                                                                           grid_max_feature_num=per))
    res = fe.process(torch.from_numpy(img).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(_u32(res.kp_keys), want) and np.array_equal(res.kp_count.cpu().numpy(), wc)
    desc, ang = oracle.orient_rbrief(flat, blur, want, wc)
    for n in range(2 * F):
        assert np.array_equal(res.desc.cpu().numpy().view(np.uint64)[n, :wc[n]], desc[n, :wc[n]])


def test_differential_fuzz_small(gpu, oracle, monkeypatch):
    """A dozen random configurations (odd sizes, borders below the patch radius, pyramids, cross-check) through
    tools/fuzz_frontend.py: every stage bit-exact against the oracle chain."""
    import os
    import runpy
    import sys
    script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_frontend.py")
    monkeypatch.setattr(sys, "argv", [script, "12", "7"])
    runpy.run_path(script, run_name="__main__")
