"""rocprofv3 target: the 8-level pyramid front-end on a resident stream (kernel table of the pyramid mode)."""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench
from visual_underwater_slam_amd.frontend import StereoOrbFrontend, ImageProcessorParams

F = int(sys.argv[1]) if len(sys.argv) > 1 else 500
dev = torch.device("cuda:0")
images = bench.make_stream(F, 0, dev)
fe = StereoOrbFrontend(bench.H, bench.W, max_frames=F, device=dev,
                       params=ImageProcessorParams(max_features=bench.KP, n_levels=8, scale_factor=1.2))
for _ in range(3):
    fe.process(images, check=False)
torch.cuda.synchronize()
