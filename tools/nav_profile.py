"""rocprofv3 target: the full graph (stereo + IMU + DVL + priors) at the configs[2] size."""
import sys, json
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from visual_underwater_slam_amd import ba_bench
n_kf = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
r = ba_bench.run_full_graph(torch.device("cuda:0"), n_kf, 25 * n_kf, 1000 if n_kf >= 1000 else 100)
print(json.dumps(r, default=str))
