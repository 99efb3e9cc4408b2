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
names = {0: "start", 1: "row_in_lds", 2: "p0_sweepA", 3: "p0_prefix", 4: "p0_sweepB", 6: "p1_sweepA", 7: "p1_prefix", 8: "p1_sweepB"}
print(json.dumps({n: t[i] - t[0] for i, n in names.items()}))
