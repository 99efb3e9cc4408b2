"""What order does PyTorch-ROCm's tensor.topk give EQUAL scores in?  (review item: tie order for small k.)
Rows with controlled ties; prints, per k, how torch's index order of each tie group relates to index-ascending."""
import torch
torch.manual_seed(0)
dev = "cuda"
L = 32760
for k in (5, 17, 32, 33, 64, 100, 128, 234, 512, 1024, 2040):
    # values: k distinct-ish levels with many ties: value = level of (i mod 37) as bf16 -> every value repeats ~885 times
    base = (torch.arange(L, device=dev) % 37).float()
    x = base.to(torch.bfloat16)[None].repeat(4, 1).contiguous()
    v, idx = x.topk(k, dim=-1)
    idx = idx[0].cpu()
    vv = v[0].float().cpu()
    asc = desc = 0
    groups = 0
    first_group = None
    for val in vv.unique():
        g = idx[vv == val]
        if len(g) < 2:
            continue
        groups += 1
        d = g[1:] - g[:-1]
        asc += bool((d > 0).all())
        desc += bool((d < 0).all())
        if first_group is None:
            first_group = g[:12].tolist()
    # which members of the cut group were selected: the smallest indices?
    cut = vv[-1]
    sel = set(idx[vv == cut].tolist())
    allcut = (x[0].float().cpu() == cut).nonzero().flatten().tolist()
    smallest = set(allcut[:len(sel)])
    largest = set(allcut[-len(sel):])
    print(f"k={k}: tie groups {groups}, index-ascending {asc}, index-descending {desc}; cut group picks smallest={sel == smallest} largest={sel == largest}; first group head {first_group}")
# gaussian maxpooled row as in the path
g = torch.Generator(device=dev).manual_seed(1)
s = torch.randn(4, L, device=dev, generator=g).to(torch.bfloat16)
s = torch.nn.functional.max_pool1d(s[None].float(), 7, 1, 3)[0].to(torch.bfloat16)
for k in (17, 32, 120, 234, 2040):
    v, idx = s.topk(k, dim=-1)
    can = torch.sort(s.float(), dim=-1, descending=True, stable=True).indices[:, :k]
    print(f"pooled gaussian k={k}: rows identical to canonical {(idx == can).all(-1).float().mean().item():.2f}; same set {(torch.sort(idx,-1).values == torch.sort(can,-1).values).all(-1).float().mean().item():.2f}")
    if k <= 32:
        print("   torch   ", idx[0].tolist())
        print("   canon   ", can[0].tolist())
