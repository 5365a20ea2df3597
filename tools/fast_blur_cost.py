#!/usr/bin/env python3
"""What the smoothing costs inside the detector launch: vus_fast_detect_adaptive with and without blur_out on the configs[1]
stream (HIP events, 1000 stereo frames)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visual_underwater_slam_amd import _lib
from visual_underwater_slam_amd.frontend import StereoOrbFrontend, ImageProcessorParams
import bench
F = 1000
dev = torch.device("cuda:0")
images = bench.make_stream(F, 0, dev)
fe = StereoOrbFrontend(720, 1280, max_frames=F, params=ImageProcessorParams(), device=dev)
fe.process(images); torch.cuda.synchronize()
st = _lib.current_stream_ptr()
p = fe.p
def run(blur):
    ts = []
    for _ in range(4):
        fe.cand_count.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.call("vus_fast_detect_adaptive", images.data_ptr(), 2 * F, 720, 1280, 1280, fe.fast_thr.data_ptr(), p.border,
                  fe.blur.data_ptr() if blur else None, fe.cand_keys.data_ptr(), p.cand_cap, fe.cand_count.data_ptr(), st)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts)
print("detect + smoothing: %.3f ms   detect only: %.3f ms" % (run(True), run(False)))
