import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyramidkv_amd as P
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
q, k, v = (torch.randn(1, 32, S, 128, device="cuda").to(torch.bfloat16) for _ in range(3))
for _ in range(3):
    P.ops.score_h2o(q, k, 8)
torch.cuda.synchronize()
