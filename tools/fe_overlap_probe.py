"""Would splitting the 1000-frame stream into chunks on two HIP streams let the front-end's stages overlap (FAST is
VALU-bound, the track matcher MFMA-bound, top-K latency-bound)?  Two StereoOrbFrontend instances, 500 frames each, on two
streams, against the same two calls on one stream.  (The chunk boundary's temporal pair is ignored: timing only.)"""
import sys, time
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torch
from visual_underwater_slam_amd import synth
from visual_underwater_slam_amd.frontend import StereoOrbFrontend, ImageProcessorParams

dev = torch.device("cuda:0")
F, H, W = 1000, 720, 1280
frames = synth.stereo_frames(0, F, H=H, W=W, xp=torch, device=dev)
for n_chunks in (1, 2, 4):
    Fc = F // n_chunks
    fes = [StereoOrbFrontend(H, W, max_frames=Fc, params=ImageProcessorParams()) for _ in range(n_chunks)]
    streams = [torch.cuda.Stream() for _ in range(min(n_chunks, 2))]
    def run(two_streams):
        for i, fe in enumerate(fes):
            st = streams[i % len(streams)] if two_streams else torch.cuda.current_stream()
            with torch.cuda.stream(st):
                fe.process(frames[i * Fc:(i + 1) * Fc], check=False)
    for two in (False, True):
        if two and n_chunks == 1:
            continue
        for _ in range(2):
            run(two); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t = time.perf_counter(); run(two); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        print(f"chunks {n_chunks}  {'two streams' if two else 'one stream '}  {1e3 * sorted(ts)[2]:.2f} ms per {F} frames")
