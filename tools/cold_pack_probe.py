"""First use of csrc/pack.hip in a fresh process: the code object's load (first launch of any kernel of the translation
unit) apart from the first radix sort (round 3: 22.9 ms of loading while the sorts were rocPRIM's, 1.5 ms with the sort of pack.hip)."""
import json, sys, time
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torch
from visual_underwater_slam_amd import _lib

torch.cuda.init(); torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()
dev = torch.device("cuda:0")
n = 2_000_000
out = {}
def phase(name, fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    out[name] = round(1e3 * (time.perf_counter() - t), 2); return r
a = torch.ones(2000, dtype=torch.int32, device=dev); b = torch.empty(2001, dtype=torch.int32, device=dev)
tot = torch.empty(1, dtype=torch.int64, device=dev)
scan = lambda: _lib.call("vus_exclusive_scan_i32", _lib.ptr(a), 2000, _lib.ptr(b), _lib.ptr(tot), _lib.current_stream_ptr())
phase("scan_first (loads the code object of pack.hip)", scan)
phase("scan_second", scan)
keys = torch.randint(0, 50000, (n,), dtype=torch.int64, device=dev)
idx = torch.empty(n, dtype=torch.int32, device=dev); uq = torch.empty(n, dtype=torch.int64, device=dev)
nu = torch.empty(1, dtype=torch.int32, device=dev)
wb = int(_lib.load().vus_pack_work_bytes(n)); work = torch.empty(wb, dtype=torch.uint8, device=dev)
k2i = lambda: _lib.call("vus_keys_to_indices", _lib.ptr(keys), n, _lib.ptr(idx), _lib.ptr(uq), _lib.ptr(nu), _lib.ptr(work), wb, _lib.current_stream_ptr())
phase("keys_to_indices_first (radix sort of 2 M keys)", k2i)
phase("keys_to_indices_second", k2i)
print(json.dumps(out))
