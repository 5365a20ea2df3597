import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
from visual_underwater_slam_amd.frontend import StereoOrbFrontend, ImageProcessorParams
F = 500
images = bench.make_stream(F, 0, torch.device("cuda:0"))
fe = StereoOrbFrontend(720, 1280, max_frames=F, params=ImageProcessorParams())
fe.process(images); torch.cuda.synchronize()
print("retried images:", int(fe.fast_retry_count.item()), "of", 2 * F, " thr min/mean/max", int(fe.fast_thr.min()), float(fe.fast_thr.float().mean()), int(fe.fast_thr.max()))
print("cand count min/mean/max", int(fe.cand_count.min()), float(fe.cand_count.float().mean()), int(fe.cand_count.max()))
