import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyramidkv_amd as P
from pyramidkv_amd import _native as N
s = torch.softmax(torch.randn(32, 32760, device="cuda") * 3, -1).to(torch.bfloat16)
for _ in range(3):
    P.ops.sort_rows(s)
buf = torch.zeros(16, dtype=torch.int64, device="cuda")
N.lib.pkv_debug_topk_trace(buf.data_ptr())
P.ops.sort_rows(s)
torch.cuda.synchronize()
N.lib.pkv_debug_topk_trace(None)
t = buf.cpu().tolist()
names = ["start", "composites", "p0_sweepA", "p0_prefix", "p0_sweepB", "p0_reload", "p1_sweepA", "p1_prefix", "p1_sweepB"]
print(json.dumps({names[i]: t[i] - t[0] for i in range(9)}))
