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
    # the asynchronous form used by bench.py (several gathers in flight, waited for later) gives the same tensor
    _, _, idx_loc = select(q[:, h0:h1], k[:, h0:h1], v[:, h0:h1])
    handles = [pdist.allgather_indices_async(idx_loc), pdist.allgather_indices_async(idx_loc + 1)]
    ok = ok and torch.equal(handles[0].wait().long(), ridx) and torch.equal(handles[1].wait().long(), ridx + 1)
    ok = ok and torch.equal(handles[0].wait().long(), ridx)            # wait() is idempotent
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


def _ada_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pyramidkv_amd import dist as pdist
    H, S, w, cap, floor = 8, 512, 8, 64, 0.2
    q, k, v = make_qkv(1, H, S, 128, "bf16", "gauss", 11)
    h0, h1 = pdist.shard_heads(H, rank, world)
    base = cap - w

    def score_sort(ql, kl):
        s = O.pool_scores(O.window_scores(ql, kl, w, "mean"), "maxpool", 7)
        srt = torch.sort(s[0], dim=-1, descending=True, stable=True)
        return srt.indices.int(), srt.values

    def budget(allv):
        # oracle budget on already-sorted rows (sorting again is the identity)
        return O.adakv_head_capacity(allv[None], base, floor, True, "canonical")[1][0]

    def gather(kl, vl, sidx, capl):
        per_head = [sidx[h, :int(capl[h])].long() for h in range(sidx.shape[0])]
        kf, vf, lens = O._flat_gather(kl, vl, per_head, w)
        hl = torch.tensor(lens, dtype=torch.int32)
        cu = torch.cat([torch.zeros(1, dtype=torch.int32), torch.cumsum(hl, 0, dtype=torch.int32)])
        return kf, vf, hl, cu

    cl = pdist.HeadShardedAdaKV(score_sort, budget, gather)
    kf, vf, hl, cu, cap_all = cl.update_kv(k[:, h0:h1], q[:, h0:h1], v[:, h0:h1])
    kr, vr, meta = O.adakv_update_kv(k, q, v, w, cap, 7, "maxpool", floor, True)
    lens = meta.head_lens.tolist()
    off0, off1 = sum(lens[:h0]), sum(lens[:h1])
    ok = hl.tolist() == lens[h0:h1] and torch.equal(kf, kr[off0:off1]) and torch.equal(vf, vr[off0:off1])
    ok = ok and [int(c) + w for c in cap_all] == lens
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_head_sharded_adakv_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_ada_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def _prefill_exchange_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pyramidkv_amd import dist as pdist
    B, H, ks = 2, 8, [11, 7, 3]
    Hl = H // world
    full = [torch.arange(B * H * k, dtype=torch.int32).view(B, H, k) * (i + 1) for i, k in enumerate(ks)]   # the unsharded truth
    xch = pdist.PrefillIndexExchange(ks, B, Hl, "cpu")
    for i in range(len(ks)):
        xch.slot(i).copy_(full[i][:, rank * Hl:(rank + 1) * Hl])          # what ops.compress(idx_out=slot) writes on the GPU
    got = xch.views(xch.gather_async())
    ret[rank] = all(torch.equal(a, b) for a, b in zip(got, full)) and len(got) == len(ks)
    dist.barrier()
    dist.destroy_process_group()


def test_one_allgather_per_prefill_world2():
    """PrefillIndexExchange: the selections of all layers travel in ONE all-gather; per-layer [B, H, k_l] views come back
    in head-major (= rank) order."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_prefill_exchange_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)
