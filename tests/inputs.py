"""Deterministic synthetic inputs shared by tests, golden generation, smoke and bench.

All generation happens on the CPU generator (bit-reproducible across machines with the same
torch build); tensors are moved to the device afterwards.
"""
import numpy as np
import torch

DTYPES = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}


def make_qkv(B, H, S, D, dtype, kind, seed, device="cpu"):
    """kind:
      gauss    N(0,1) fp32 -> cast (SURVEY section 8d 'tier B')
      lattice  values in {-4..4}/4: every q.k dot product is exact in fp32 in any order ('tier A')
      planted  gauss background + a few hundred 'heavy hitter' keys whose logits are
               well separated, so selection is robust to 1-ulp score noise
      sink     what a real prompt looks like to this path (round 6): logits with standard deviation 6 (q, k ~ N(0, 6^(1/2)))
               and an attention-sink key at position 0 that every one of the last 64 queries scores at exactly +40: next
               to it exp(x - 40) underflows for all but a handful of keys - in fp16 most pooled scores are exactly 0 and
               the k-th largest is tied thousands of times (bf16 keeps fp32's exponent range: a peaked row without ties)
    """
    g = torch.Generator().manual_seed(seed)
    dt = DTYPES[dtype] if isinstance(dtype, str) else dtype
    if kind == "gauss":
        q, k, v = (torch.randn(B, H, S, D, generator=g) for _ in range(3))
    elif kind == "lattice":
        q, k, v = (torch.randint(-4, 5, (B, H, S, D), generator=g).float() / 4 for _ in range(3))
    elif kind == "planted":
        # tiny background + 64 heavy-hitter keys aligned with a direction u shared by the window
        # queries; hitter logits are ~0.1 apart (>> 1 ulp of the score dtype after softmax)
        q = torch.randn(B, H, S, D, generator=g) * 0.02
        k = torch.randn(B, H, S, D, generator=g) * 0.02
        v = torch.randn(B, H, S, D, generator=g)
        n_hit = 64
        for b in range(B):
            for h in range(H):
                pos = (torch.randperm((S - 128) // 16, generator=g)[:n_hit] * 16 + 8) if S >= 2048 else torch.arange(0)
                u = torch.randn(D, generator=g)
                u = u / u.norm()
                q[b, h, -64:] += 6.0 * u
                strength = torch.linspace(2.0, 14.0, len(pos))
                k[b, h, pos] += strength[:, None] * u
    elif kind == "sink":
        a = 6.0 ** 0.5
        q = torch.randn(B, H, S, D, generator=g) * a
        k = torch.randn(B, H, S, D, generator=g) * a
        v = torch.randn(B, H, S, D, generator=g)
        u = torch.randn(B, H, 1, D, generator=g)
        u = u / u.norm(dim=-1, keepdim=True)
        nq = min(64, S)
        qw = q[:, :, -nq:]
        q[:, :, -nq:] = qw - (qw * u).sum(-1, keepdim=True) * u + 4.0 * u      # component along u: exactly 4
        k[:, :, 0:1] = (40.0 * D ** 0.5 / 4.0) * u                             # q . k0 / sqrt(D) = 40
    else:
        raise ValueError(kind)
    return q.to(dt).to(device), k.to(dt).to(device), v.to(dt).to(device)


def bits(t: torch.Tensor) -> np.ndarray:
    """Bit pattern of a tensor as a numpy integer array (bf16/fp16 -> uint16, fp32 -> uint32)."""
    t = t.detach().cpu().contiguous()
    if t.dtype in (torch.bfloat16, torch.float16):
        return t.view(torch.int16).numpy().view(np.uint16)
    if t.dtype == torch.float32:
        return t.view(torch.int32).numpy().view(np.uint32)
    return t.numpy()


def from_bits(a: np.ndarray, dtype) -> torch.Tensor:
    dt = DTYPES[dtype] if isinstance(dtype, str) else dtype
    if dt in (torch.bfloat16, torch.float16):
        return torch.from_numpy(a.view(np.int16).copy()).view(dt)
    return torch.from_numpy(a.view(np.int32).copy()).view(dt)


def checksum(*tensors) -> int:
    """Order-sensitive 63-bit checksum of bit patterns."""
    acc = 0
    for t in tensors:
        a = bits(t).astype(np.uint64).ravel()
        wts = (np.arange(a.size, dtype=np.uint64) % np.uint64(65521)) + np.uint64(1)
        acc = (acc * 1000003 + int((a * wts).sum() % np.uint64(2**61 - 1))) % (2**63 - 1)
    return acc
