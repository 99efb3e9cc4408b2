"""Phase stamps (shader clock) of topk_kernel row 0 inside pkv_compress (chunk-maxima prefilter path)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyramidkv_amd as P
from pyramidkv_amd import _native as N
res = {}
q, k, v = (torch.randn(1, 32, 32768, 128, device="cuda").to(torch.bfloat16) for _ in range(3))
for kk in (17, 120, 234):
    buf = torch.zeros(16, dtype=torch.int64, device="cuda")
    for _ in range(3):
        P.ops.compress(q, k, v, 8, kk, "maxpool", 7)
    N.lib.pkv_debug_topk_trace(buf.data_ptr())
    P.ops.compress(q, k, v, 8, kk, "maxpool", 7)
    torch.cuda.synchronize()
    N.lib.pkv_debug_topk_trace(None)
    t = buf.cpu().tolist()
    st = [t[i] - t[0] if t[i] else None for i in range(8)]
    res[f"k{kk}"] = {"stamps_rel": st, "C": t[15], "total": t[6] - t[0],
                     "finalize_rel": [t[9] - t[8], t[10] - t[8], t[11] - t[8], t[12] - t[8]]}
print(json.dumps(res, indent=1))
