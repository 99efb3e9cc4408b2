import torch, time, sys
sys.path.insert(0, ".")
import pyramidkv_amd as P
from pyramidkv_amd import _native as N
q, k, v = (torch.randn(1, 32, 32768, 128, device="cuda") for _ in range(3))
for cap in (128, 2048):
    for _ in range(3): P.ops.compress(q, k, v, 8, cap - 8, "maxpool", 7)
    torch.cuda.synchronize(); N.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(10): P.ops.compress(q, k, v, 8, cap - 8, "maxpool", 7)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    r = N.prof_read(); N.prof_enable(False)
    print("fp32 S=32768 H=32 cap", cap, "update_kv ms", round(dt * 1e3, 3), {kk: round(1e3 * vv[0] / vv[1], 1) for kk, vv in r.items() if vv[1]})
s = torch.rand(32, 32760, device="cuda")
for kk in (120, 2040):
    for _ in range(3): P.ops.topk(s, kk)
    torch.cuda.synchronize(); N.prof_enable(True)
    for _ in range(10): P.ops.topk(s, kk)
    torch.cuda.synchronize(); r = N.prof_read(); N.prof_enable(False)
    print("topk alone uniform scores k", kk, {a: round(1e3 * b[0] / b[1], 1) for a, b in r.items() if b[1]})
