import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyramidkv_amd as P
out = {}
for L in (4088, 8184, 32760):
    s = torch.softmax(torch.randn(32, L, device="cuda") * 3, -1).to(torch.bfloat16)
    for wv in (True, False):
        for _ in range(5):
            P.ops.sort_rows(s, want_values=wv)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            P.ops.sort_rows(s, want_values=wv)
        b.record(); torch.cuda.synchronize()
        out[f"L{L}_values{int(wv)}"] = round(a.elapsed_time(b) * 1e3 / 50, 1)
print(json.dumps(out))
