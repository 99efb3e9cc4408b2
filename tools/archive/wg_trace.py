"""Per-workgroup start/end wall clock (100 MHz) for the four kernels of one update_kv call."""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyramidkv_amd as P
from pyramidkv_amd import _native as N
res = {}
for B, cap in ((1, 128), (1, 2048), (8, 128)):
    S = 32768
    q, k, v = (torch.randn(B, 32, S, 128, device="cuda").to(torch.bfloat16) for _ in range(3))
    for _ in range(3):
        P.ops.compress(q, k, v, 8, cap - 8, "maxpool", 7)
    buf = torch.zeros(2 * 262144, dtype=torch.int64, device="cuda")
    N.lib.pkv_debug_wg_trace(buf.data_ptr())
    P.ops.compress(q, k, v, 8, cap - 8, "maxpool", 7)
    torch.cuda.synchronize()
    N.lib.pkv_debug_wg_trace(None)
    t = buf.cpu().numpy().reshape(-1, 2)
    out = {}
    t0 = None
    for name, off in (("logits", 0), ("finalize", 65536), ("topk", 131072), ("gather", 196608)):
        r = t[off:off + 65536]
        r = r[r[:, 1] > 0]
        if t0 is None:
            t0 = r[:, 0].min()
        st, en = (r[:, 0] - t0) / 100.0, (r[:, 1] - t0) / 100.0     # microseconds
        life = en - st
        out[name] = dict(wgs=int(len(r)), first_start_us=round(float(st.min()), 2), last_start_us=round(float(st.max()), 2),
                         last_end_us=round(float(en.max()), 2), span_us=round(float(en.max() - st.min()), 2),
                         wg_life_us=dict(min=round(float(life.min()), 2), median=round(float(np.median(life)), 2),
                                         p90=round(float(np.percentile(life, 90)), 2), max=round(float(life.max()), 2)),
                         start_percentiles_us=[round(float(x), 2) for x in np.percentile(st - st.min(), [10, 50, 90, 99])])
    res[f"B{B}_cap{cap}"] = out
    del q, k, v
print(json.dumps(res, indent=1))
