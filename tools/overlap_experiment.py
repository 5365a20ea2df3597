"""Experiment: detect/select/describe of sub-batches on two HIP streams (FAST of batch b+1 overlapping the
descriptor kernel of batch b) vs the single-stream order.  Prints ms per 1000 frames."""
import sys, time
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench
from visual_underwater_slam_amd import _lib
from visual_underwater_slam_amd.frontend import StereoOrbFrontend, ImageProcessorParams

F = 1000
dev = torch.device("cuda:0")
images = bench.make_stream(F, 0, dev)
fe = StereoOrbFrontend(bench.H, bench.W, max_frames=F, device=dev, params=ImageProcessorParams(max_features=bench.KP))
p, H, W, K = fe.p, fe.H, fe.W, fe.p.max_features
ptr = _lib.ptr


def phase_a(lo, hi, st):
    """detect + top-K + describe for images [lo, hi) on the current stream"""
    n = hi - lo
    img = images.view(-1, H, W)[lo:hi]
    fe.cand_count[lo:hi].zero_()
    _lib.call("vus_fast_detect", ptr(img), n, H, W, W, p.fast_threshold, p.border, ptr(fe.blur[lo:hi]),
              ptr(fe.cand_keys[lo:hi]), p.cand_cap, ptr(fe.cand_count[lo:hi]), st)
    _lib.call("vus_select_topk", ptr(fe.cand_keys[lo:hi]), ptr(fe.cand_count[lo:hi]), n, p.cand_cap, K,
              ptr(fe.kp_keys[lo:hi]), ptr(fe.kp_count[lo:hi]), st)
    _lib.call("vus_orient_rbrief", ptr(img), ptr(fe.blur[lo:hi]), n, H, W, W, ptr(fe.kp_keys[lo:hi]),
              ptr(fe.kp_count[lo:hi]), K, ptr(fe.desc[lo:hi]), ptr(fe.angle[lo:hi]), st)


def phase_b(st):
    _lib.call("vus_hamming_match", ptr(fe.desc), ptr(fe.kp_keys), ptr(fe.kp_count), K, H, W, ptr(fe.stereo_q),
              ptr(fe.stereo_t), F, p.stereo_threshold, p.min_disparity, p.max_disparity, p.stereo_max_distance,
              ptr(fe.match_idx), ptr(fe.match_dist), st)
    _lib.call("vus_hamming_match", ptr(fe.desc), ptr(fe.kp_keys), ptr(fe.kp_count), K, H, W, ptr(fe.track_q),
              ptr(fe.track_t), F - 1, -1, 0, 0, p.track_max_distance, ptr(fe.match_idx[fe.max_frames:]),
              ptr(fe.match_dist[fe.max_frames:]), st)


def run(n_sub, n_streams):
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    main = torch.cuda.current_stream()
    edges = [2 * F * b // n_sub for b in range(n_sub + 1)]
    for s in streams:
        s.wait_stream(main)
    for b in range(n_sub):
        s = streams[b % n_streams]
        with torch.cuda.stream(s):
            phase_a(edges[b], edges[b + 1], _lib.current_stream_ptr())
    for s in streams:
        main.wait_stream(s)
    phase_b(_lib.current_stream_ptr())


for n_sub, n_streams in [(1, 1), (4, 1), (1, 1), (2, 1), (4, 1), (8, 1), (4, 2), (1, 1)]:
    for _ in range(2):
        run(n_sub, n_streams)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        run(n_sub, n_streams)
    torch.cuda.synchronize()
    print(f"sub-batches {n_sub:2d} streams {n_streams}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per {F} frames")
