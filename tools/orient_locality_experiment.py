#!/usr/bin/env python3
"""Experiment: how much faster is orient_rbrief when each image's keypoints are processed in a
spatially coherent order (sorted by 64x64 tile) instead of score order?  (Upper bound for an
order-indirection inside the kernel.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visual_underwater_slam_amd import _lib
from visual_underwater_slam_amd.frontend import StereoOrbFrontend, ImageProcessorParams
import bench
F = 300
dev = torch.device("cuda:0")
images = bench.make_stream(F, 0, dev)
fe = StereoOrbFrontend(720, 1280, max_frames=F, params=ImageProcessorParams(), device=dev)
fe.process(images); torch.cuda.synchronize()
def time_orient(keys):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        ev0.record()
        _lib.call("vus_orient_rbrief", images.data_ptr(), fe.blur.data_ptr(), 2 * F, 720, 1280, 1280, keys.data_ptr(),
                  fe.kp_count.data_ptr(), 2000, fe.desc.data_ptr(), fe.angle.data_ptr(), _lib.current_stream_ptr())
        ev1.record(); torch.cuda.synchronize()
    return ev0.elapsed_time(ev1)
k = fe.kp_keys.clone()
print("score order      : %.3f ms" % time_orient(k))
pos = (k.long() & 0xFFFFFF); y = pos // 1280; x = pos % 1280
for tile in (32, 64, 128):
    tid = (y // tile) * 64 + (x // tile)
    order = torch.argsort(tid, dim=1, stable=True)
    ks = torch.gather(k, 1, order).contiguous()
    print("tile %3d order   : %.3f ms" % (tile, time_orient(ks)))
order = torch.argsort(pos, dim=1)
print("raster order     : %.3f ms" % time_orient(torch.gather(k, 1, order).contiguous()))
