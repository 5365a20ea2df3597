"""PCIe-inclusive rate: frames start in pinned host memory; chunks are uploaded on a copy stream while the previous
chunk is processed (double buffering).  Not the bench metric (that one has its inputs resident in HBM)."""
import sys, time
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench
from visual_underwater_slam_amd.frontend import StereoOrbFrontend, ImageProcessorParams

dev = torch.device("cuda:0")
CH = int(sys.argv[1]) if len(sys.argv) > 1 else 64          # frames per chunk
N_CH = 16
src = bench.make_stream(CH, 0, dev).cpu().pin_memory()      # one chunk of frames on the host, re-used for every upload
fe = StereoOrbFrontend(bench.H, bench.W, max_frames=CH, device=dev, params=ImageProcessorParams(max_features=bench.KP))
bufs = [torch.empty_like(src, device=dev) for _ in range(2)]
copy_stream = torch.cuda.Stream()
ready = [torch.cuda.Event() for _ in range(2)]
done = [torch.cuda.Event() for _ in range(2)]


def run():
    for c in range(N_CH):
        b = c & 1
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(done[b])                  # the previous user of this buffer has finished
            bufs[b].copy_(src, non_blocking=True)
            ready[b].record(copy_stream)
        torch.cuda.current_stream().wait_event(ready[b])
        fe.process(bufs[b], check=False)
        done[b].record()


for e in done:
    e.record()
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
run(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
gb = N_CH * src.numel() / 1e9
print(f"chunk {CH} frames: {N_CH * CH / dt:.0f} stereo frames/s with the upload in the loop ({gb / dt:.1f} GB/s host->device)")
t0 = time.perf_counter()
for _ in range(N_CH):
    bufs[0].copy_(src, non_blocking=True)
torch.cuda.synchronize()
print(f"upload alone: {gb / (time.perf_counter() - t0):.1f} GB/s")
