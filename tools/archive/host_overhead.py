"""Host-side cost of one ops.compress call (Python + ctypes + 3 torch.empty + 4 launches): tiny S, so the GPU never limits."""
import json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyramidkv_amd as P
q, k, v = (torch.randn(1, 32, 512, 128, device="cuda").to(torch.bfloat16) for _ in range(3))
for _ in range(50):
    P.ops.compress(q, k, v, 8, 56, "maxpool", 7, return_indices=True)
torch.cuda.synchronize()
n = 3000
t0 = time.perf_counter()
for _ in range(n):
    P.ops.compress(q, k, v, 8, 56, "maxpool", 7, return_indices=True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(json.dumps({"host_us_per_call": round((t1 - t0) / n * 1e6, 2), "incl_drain_us_per_call": round((t2 - t0) / n * 1e6, 2)}))
