"""First-GPU-session probe: device facts, achievable HBM bandwidth, and the
reference's eager pipeline executed by PyTorch-ROCm on this chip (the same-chip comparator)."""
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pkv_oracle as O  # noqa: E402

out = {}
p = torch.cuda.get_device_properties(0)
out["device"] = dict(name=p.name, cus=p.multi_processor_count, mem_gb=round(p.total_memory / 2**30, 1),
                     gcn=getattr(p, "gcnArchName", ""), l2_mb=getattr(p, "L2_cache_size", 0) / 2**20)
out["host"] = dict(cpus=os.cpu_count(), reference_present=os.path.exists("/root/reference"))
try:
    out["host"]["lscpu"] = [l for l in subprocess.run(["lscpu"], capture_output=True, text=True).stdout.splitlines()
                            if l.startswith(("Model name", "CPU(s):", "Thread", "Socket"))]
except Exception as ex:  # noqa: BLE001
    out["host"]["lscpu"] = str(ex)


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


x = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
y = torch.empty_like(x)
ms = timeit(lambda: y.copy_(x))
out["hbm_copy_GBps_1GiB"] = round(2 * x.numel() / ms / 1e6, 1)
xs = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
ms = timeit(lambda: xs.sum())
out["hbm_read_GBps_sum_1GiB"] = round(xs.numel() * 4 / ms / 1e6, 1)
del x, y, xs

# reference pipeline (oracle restatement = same ATen ops) on the device, eager
res = {}
for S, cap in ((8192, 128), (32768, 128), (32768, 2048)):
    q, k, v = (torch.randn(1, 32, S, 128, device="cuda").to(torch.bfloat16) for _ in range(3))
    ms = timeit(lambda: O.snapkv_update_kv(k, q, v, 8, cap, 7, "maxpool", topk_mode="reference"), n=5, warm=2)
    res[f"snapkv_S{S}_cap{cap}_ms"] = round(ms, 3)
out["torch_rocm_eager_reference"] = res
print(json.dumps(out, indent=1))
