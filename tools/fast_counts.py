#!/usr/bin/env python3
"""How many strips / pixels reach the second and third stage of the FAST tile kernel on the configs[1] stream (a counting
build of the library: tools/ab/build_frontend_variant.sh cnt -DVUS_FAST_DEBUG_COUNT, VUS_HIP_LIB=tools/ab/libvus_fe_cnt.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visual_underwater_slam_amd import _lib
from visual_underwater_slam_amd.frontend import StereoOrbFrontend, ImageProcessorParams
import bench
F = 100
dev = torch.device("cuda:0")
images = bench.make_stream(F, 0, dev)
fe = StereoOrbFrontend(720, 1280, max_frames=F, params=ImageProcessorParams(), device=dev)
lib = _lib.load()
out = (ctypes.c_ulonglong * 4)()
fe.process(images); torch.cuda.synchronize()
lib.vus_debug_fast_counters(out, 1)
fe.process(images); torch.cuda.synchronize()
lib.vus_debug_fast_counters(out, 1)
tiles, strips, pixels = out[0], out[1], out[2]
print(f"tiles {tiles}  strips listed per tile {strips / tiles:.1f} of 884 ({100 * strips / tiles / 884:.1f} %)  "
      f"pixels listed per tile {pixels / tiles:.1f} of 3536 ({100 * pixels / tiles / 3536:.2f} %)")
print("thresholds of the first images:", fe.fast_thr[:8].tolist())
