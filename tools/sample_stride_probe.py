import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
from visual_underwater_slam_amd.frontend import StereoOrbFrontend, ImageProcessorParams
F = 500
images = bench.make_stream(F, 0, torch.device("cuda:0"))
for stride in (32, 48, 64, 96):
    fe = StereoOrbFrontend(720, 1280, max_frames=F, params=ImageProcessorParams(fast_sample_stride=stride))
    fe.process(images); torch.cuda.synchronize()
    t = bench.StageTimer(); fe.stage_hook = t
    for _ in range(5):
        fe.process(images, check=False)
    torch.cuda.synchronize()
    ms = t.stage_ms()
    print(f"stride {stride}: retried {int(fe.fast_retry_count.item())} of {2*F}, thr mean {float(fe.fast_thr.float().mean()):.1f}, "
          f"estimate {ms['fast_threshold']*2:.3f} + detect {ms['fast_detect']*2:.3f} = {(ms['fast_threshold']+ms['fast_detect'])*2:.3f} ms per 1000 frames")
    del fe
