"""gather_kernel rows-per-lane (PKV_GATHER_RPT) at budget 2048, B = 1 and 8: hipEvent time of the gather alone, launched
back to back on the same K/V.  CAUTION: at B = 1 the 134 MB of K/V stay in the 256 MB Infinity Cache between launches, so
these numbers flatter the kernel (16 rows per lane looked 10 % faster here and made no difference inside update_kv, where
the rows come from HBM: bench.py roofline_kernels.gather_cap2048_B1, PKV_GATHER_RPT=8 / 16: 12.8-12.9 / 12.9-13.1 us)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import torch
    import pyramidkv_amd as P
    from pyramidkv_amd import _native as N
    out = {}
    S, H, w, cap = 32768, 32, 8, 2048
    for B in (1, 8):
        k, v = (torch.randn(B, H, S, 128, device="cuda").to(torch.bfloat16) for _ in range(2))
        idx = torch.stack([torch.randperm(S - w, device="cuda")[:cap - w] for _ in range(B * H)]).view(B, H, cap - w).int()
        for _ in range(5):
            P.ops.gather_compact(k, v, idx, w)
        torch.cuda.synchronize(); N.prof_enable(True)
        for _ in range(30):
            P.ops.gather_compact(k, v, idx, w)
        torch.cuda.synchronize(); r = N.prof_read(); N.prof_enable(False)
        us = 1e3 * r["gather"][0] / r["gather"][1]
        out["B%d" % B] = [round(us, 2), round(B * H * cap * 256 * 2 * 2 / us / 1e3 / 8000, 3)]
        del k, v
    print(json.dumps(out))
    sys.exit(0)
for rep in range(2):
    for rpt in ("0", "4", "8", "16"):
        r = subprocess.run([sys.executable, __file__, "--one"], env=dict(os.environ, PKV_GATHER_RPT=rpt), capture_output=True, text=True)
        print("rpt", rpt, r.stdout.strip().splitlines()[-1] if r.returncode == 0 else r.stderr[-300:], flush=True)
