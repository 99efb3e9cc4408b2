"""The north-star gather (SnapKV budget 2048, S = 32768) at B = 1 and B = 8, a few launches each: the workload of the
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes whose per-launch traffic tools/pmc_summary.py files under
gather_cap2048_B1 / _B8 (told apart by the launch grid)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyramidkv_amd as P
S, H, w, cap = 32768, 32, 8, 2048
for B in (1, 8):
    q, k, v = (torch.randn(B, H, S, 128, device="cuda").to(torch.bfloat16) for _ in range(3))
    idx = torch.stack([torch.randperm(S - w, device="cuda")[:cap - w] for _ in range(B * H)]).view(B, H, cap - w).int()
    for _ in range(4):
        P.ops.gather_compact(k, v, idx, w)
    torch.cuda.synchronize()
    del q, k, v
