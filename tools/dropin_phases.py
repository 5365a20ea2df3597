"""Phases of the drop-in call gtsam.LevenbergMarquardtOptimizer(graph, initial, params).optimize() at configs[2]
(the `dropin` object of the bench line, alone).  usage: python tools/dropin_phases.py [repeats]"""
import json, sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from visual_underwater_slam_amd import ba_bench
import torch
r = ba_bench.run(torch.device("cuda:0"), reps=int(sys.argv[1]) if len(sys.argv) > 1 else 3)
print(json.dumps({"lm_s": r["value"], "from_arrays_s": r["value_cold"], "structure_setup_s": r["structure_setup_s"],
                  "dropin_s": r["dropin"]["value"], "phase_ms": r["dropin"]["phase_ms"]}))
