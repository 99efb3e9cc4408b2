"""World-size-2 test of the head-sharded path on CPU (gloo): shard selection, the single all-gather of
indices and its head-major layout.  The per-shard selection is stood in for by the oracle here (no GPU
in this container); on the GPU box the same wrapper runs the HIP path over RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from inputs import make_qkv
from oracle import pkv_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pyramidkv_amd import dist as pdist
    B, H, S, w, cap = 2, 8, 512, 8, 64
    q, k, v = make_qkv(B, H, S, 128, "bf16", "gauss", 3)
    h0, h1 = pdist.shard_heads(H, rank, world)

    def select(ql, kl, vl):
        kc, vc, idx = O.snapkv_update_kv(kl, ql, vl, w, cap, 7, "maxpool", return_indices=True)
        return kc, vc, idx.int()

    cl = pdist.HeadShardedCluster(select)
    kc, vc, idx_all = cl.update_kv(k[:, h0:h1], q[:, h0:h1], v[:, h0:h1])
    kr, vr, ridx = O.snapkv_update_kv(k, q, v, w, cap, 7, "maxpool", return_indices=True)
    ok = torch.equal(idx_all.long(), ridx) and torch.equal(kc, kr[:, h0:h1]) and torch.equal(vc, vr[:, h0:h1])
    ret[rank] = bool(ok) and idx_all.shape == (B, H, cap - w)
    dist.barrier()
    dist.destroy_process_group()


def test_head_sharded_allgather_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_shard_heads_partition():
    from pyramidkv_amd import dist as pdist
    for H, N in ((32, 1), (32, 2), (32, 4), (32, 8)):
        spans = [pdist.shard_heads(H, r, N) for r in range(N)]
        assert spans[0][0] == 0 and spans[-1][1] == H
        assert all(spans[i][1] == spans[i + 1][0] for i in range(N - 1))
    with pytest.raises(ValueError):
        pdist.shard_heads(30, 0, 4)
