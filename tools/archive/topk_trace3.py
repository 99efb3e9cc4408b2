"""Phase stamps (shader clock) of topk_kernel row 0 inside the fused update_kv (chunk maxima from finalize), per layer
budget.  Needs the -DPKV_DEBUG build:  PKV_LIB=pyramidkv_amd/libpkv_debug.so python tools/topk_trace3.py"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyramidkv_amd as P
from pyramidkv_amd import _native as N
assert N.lib.pkv_debug_build() == 1, "load the debug build through PKV_LIB"
res = {}
q, kk, v = (torch.randn(1, 32, 32768, 128, device="cuda").to(torch.bfloat16) for _ in range(3))
for k in (17, 66, 120, 234, 512, 2040):
    buf = torch.zeros(16, dtype=torch.int64, device="cuda")
    for _ in range(3):
        P.ops.compress(q, kk, v, 8, k, "maxpool", 7)
    N.lib.pkv_debug_topk_trace(buf.data_ptr())
    P.ops.compress(q, kk, v, 8, k, "maxpool", 7)
    torch.cuda.synchronize()
    N.lib.pkv_debug_topk_trace(None)
    t = buf.cpu().tolist()
    res["k%d" % k] = {"stamps_rel": [x - t[0] if x else None for x in t[:8]], "C": t[15], "total": t[6] - t[0], "s13": t[13] - t[0] if t[13] else None,
                      "finalize_rel": [t[9] - t[8], t[10] - t[8], t[11] - t[8], t[12] - t[8]]}
print(json.dumps(res, indent=1))
