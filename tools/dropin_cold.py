"""The drop-in call as batch.py makes it: ONE gtsam.LevenbergMarquardtOptimizer(...).optimize() in a fresh process
(batch.py:337).  Times the first call (library load, code objects, first allocations included) and a second one."""
import json, sys, time
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import numpy as np
t_imp = time.perf_counter()
import torch
from visual_underwater_slam_amd import synth, gtsam, ba_bench
t_imp = time.perf_counter() - t_imp
n_kf = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
s = synth.ba_sequence(n_kf, 25 * n_kf, 1000)
graph, initial = ba_bench.build_graph(s, len(s["points_gt"]), n_kf)
t0 = time.perf_counter()
torch.cuda.init(); torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()
t_ctx = time.perf_counter() - t0
import os
os.environ["VUS_PROFILE_BOUNDARY"] = "1"       # synchronising phase marks (adds < 1 ms)
ts, phases = [], []
for _ in range(3):
    t = time.perf_counter()
    o = gtsam.LevenbergMarquardtOptimizer(graph, initial, gtsam.LevenbergMarquardtParams())
    o.optimize()
    ts.append(time.perf_counter() - t)
    phases.append(getattr(o.report(), "boundary_ms", None))
print(json.dumps({"first_call_phase_ms": phases[0], "second_call_phase_ms": phases[1], "keyframes": n_kf, "stereo_factors": len(s["obs_pose"]), "import_s": round(t_imp, 2), "gpu_context_s": round(t_ctx, 2),
                  "first_call_s": round(ts[0], 4), "second_call_s": round(ts[1], 4), "third_call_s": round(ts[2], 4)}))
