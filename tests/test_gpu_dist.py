"""Multi-process tests of the head-sharded path ON THE GPU (round-2 VERDICT item 5):

  * world size 2, two processes on ONE device, gloo process group: every rank runs the HIP path on its head shard
    (``HeadShardedCluster`` / the HIP-wired ``HeadShardedAdaKV``); the concatenation over the ranks equals the unsharded
    HIP result AND the oracle;
  * the C-ABI collective ``pkv_allgather_indices`` over a real RCCL communicator (nranks = 1: this box has one GPU).

RCCL refuses two ranks on one device, so the world-2 runs use gloo; what they exercise is everything around the
collective: sharding, per-rank HIP kernels, the exchange layout, the global Ada-SnapKV budget from exchanged lists.
"""
import ctypes
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from inputs import make_qkv
from oracle import pkv_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _snap_worker(rank, world, port, ret):
    _init(rank, world, port)
    import pyramidkv_amd as P
    from pyramidkv_amd import dist as pdist
    B, H, S, w, cap = 2, 8, 4096, 8, 128
    q, k, v = make_qkv(B, H, S, 128, "bf16", "gauss", 3)
    h0, h1 = pdist.shard_heads(H, rank, world)
    qd, kd, vd = (t[:, h0:h1].contiguous().to(DEV) for t in (q, k, v))

    def select(ql, kl, vl):
        return P.ops.compress(ql, kl, vl, w, cap - w, "maxpool", 7, return_indices=True)

    kc, vc, idx_all = pdist.HeadShardedCluster(select).update_kv(kd, qd, vd)
    # unsharded HIP run on this rank's device and the oracle on the CPU
    kc_u, vc_u, idx_u = P.ops.compress(q.to(DEV), k.to(DEV), v.to(DEV), w, cap - w, "maxpool", 7, return_indices=True)
    kr, vr, ridx = O.snapkv_update_kv(k, q, v, w, cap, 7, "maxpool", return_indices=True)
    ok = idx_all.shape == (B, H, cap - w)
    ok = ok and torch.equal(idx_all, idx_u) and torch.equal(idx_all.cpu().long(), ridx)
    ok = ok and torch.equal(kc, kc_u[:, h0:h1]) and torch.equal(vc, vc_u[:, h0:h1])
    ok = ok and torch.equal(kc.cpu(), kr[:, h0:h1]) and torch.equal(vc.cpu(), vr[:, h0:h1])
    # the asynchronous form bench.py uses
    hnd = pdist.allgather_indices_async(select(qd, kd, vd)[2])
    ok = ok and torch.equal(hnd.wait(), idx_u)
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def _ada_worker(rank, world, port, ret):
    _init(rank, world, port)
    import pyramidkv_amd as P
    from pyramidkv_amd import dist as pdist
    H, S, w, cap, floor = 8, 4096, 8, 64, 0.2
    q, k, v = make_qkv(1, H, S, 128, "bf16", "gauss", 11)
    h0, h1 = pdist.shard_heads(H, rank, world)
    qd, kd, vd = (t[:, h0:h1].contiguous().to(DEV) for t in (q, k, v))
    cl = pdist.hip_head_sharded_adakv(H, w, 7, "maxpool", cap, floor, True)
    kf, vf, hl, cu, cap_all = cl.update_kv(kd, qd, vd)
    # unsharded HIP cluster and the oracle
    full = P.AdaKVCluster(window_size=w, kernel_size=7, pooling="maxpool", max_capacity_prompt=cap, floor=floor, normalize=True)
    kf_u, vf_u = full.update_kv(k.to(DEV), q.to(DEV), v.to(DEV))
    kr, vr, meta = O.adakv_update_kv(k, q, v, w, cap, 7, "maxpool", floor, True)
    lens = meta.head_lens.tolist()
    off0, off1 = sum(lens[:h0]), sum(lens[:h1])
    ok = [int(c) + w for c in cap_all.cpu()] == lens == full.head_lens.cpu().tolist()
    ok = ok and hl.cpu().tolist() == lens[h0:h1]
    ok = ok and torch.equal(kf, kf_u[off0:off1]) and torch.equal(vf, vf_u[off0:off1])
    ok = ok and torch.equal(kf.cpu(), kr[off0:off1]) and torch.equal(vf.cpu(), vr[off0:off1])
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def _config4_worker(rank, world, port, q, k, v, ks, oracle_idx, ret):
    """BASELINE config 4 on ONE device: rank r owns heads [4r, 4r+4) of one 32k sequence, all 32 PyramidKV budget-128
    layer budgets go through PrefillIndexExchange (one all-gather per prefill)."""
    _init(rank, world, port)
    import pyramidkv_amd as P
    from pyramidkv_amd import dist as pdist
    H, w = q.shape[1], 8
    h0, h1 = pdist.shard_heads(H, rank, world)
    qd, kd, vd = (t[:, h0:h1].contiguous().to(DEV) for t in (q, k, v))
    xch = pdist.PrefillIndexExchange(ks, 1, h1 - h0, torch.device(DEV, 0))
    outs = []
    for layer, kk in enumerate(ks):
        kc, vc, _ = P.ops.compress(qd, kd, vd, w, kk, "maxpool", 7, idx_out=xch.slot(layer))
        outs.append((kc, vc))
    gathered = xch.views(xch.gather_async())
    ok = len(gathered) == len(ks)
    bad = []
    for layer, kk in enumerate(ks):
        g = gathered[layer]
        ok_l = g.shape == (1, H, kk) and torch.equal(g.cpu().long(), oracle_idx[layer])        # every rank holds ALL heads' indices == the oracle's
        if ok_l:                                                                                 # this rank's compacted K/V == the oracle's gather of its heads
            gi = oracle_idx[layer][:, h0:h1]
            kr, vr = O.gather_compact(k[:, h0:h1], v[:, h0:h1], gi, w)
            ok_l = torch.equal(outs[layer][0].cpu(), kr) and torch.equal(outs[layer][1].cpu(), vr)
        if not ok_l:
            bad.append(layer)
    if rank == 0:      # the unsharded HIP path on the whole [1, 32, S, 128] tensors
        qf, kf, vf = q.to(DEV), k.to(DEV), v.to(DEV)
        for layer in (0, 13, 31):
            _, _, iu = P.ops.compress(qf, kf, vf, w, ks[layer], "maxpool", 7, return_indices=True)
            if not torch.equal(iu, gathered[layer]):
                bad.append(("unsharded", layer))
    ret[rank] = (bool(ok and not bad), bad)
    dist.barrier()
    dist.destroy_process_group()


def test_config4_world8_one_gpu():
    """BASELINE config 4 exactly as north_star partitions it - PyramidKV budget 128, S = 32768, 32 heads sharded 4 per rank
    over EIGHT ranks, one exchange of the selected indices per prefill - with the eight ranks on this box's one GPU (gloo
    carries the collective; RCCL refuses two ranks per device): gathered indices == unsharded HIP == oracle for all 32 layer
    budgets, every rank's compacted K/V == the oracle's gather of its heads."""
    import pyramidkv_amd as P
    world, H, S, w, cap = 8, 32, 32768, 8, 128
    q, k, v = make_qkv(1, H, S, 128, "bf16", "gauss", 4400)
    ks = []
    for layer in range(32):
        branch, kk = O.pyramid_budget(cap, w, 32, layer, S)
        assert branch == "pyramid"
        ks.append(kk)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    s = O.pool_scores(O.window_scores(q, k, w), "maxpool", 7)
    order = O.topk_canonical(s, max(ks))                       # (value desc, index asc): a prefix is the top-k of every smaller k
    oracle_idx = [order[..., :kk].contiguous() for kk in ks]
    for t in (q, k, v):
        t.share_memory_()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_config4_worker, args=(world, _free_port(), q, k, v, ks, oracle_idx, ret), nprocs=world, join=True)
    assert all(ret.get(r, (False,))[0] for r in range(world)), dict(ret)


@pytest.mark.parametrize("worker", [_snap_worker, _ada_worker], ids=["snapkv", "adakv"])
def test_head_sharded_world2_hip_path(worker):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_cabi_allgather_indices_over_rccl_nranks1():
    """include/pkv.h pkv_allgather_indices on a REAL RCCL communicator (created here through the RCCL the process already
    holds - PyTorch's): nranks = 1, B > 1 path included (gather + regroup kernel)."""
    import pyramidkv_amd as P
    N = P._native
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    if not os.path.exists(path):
        path = "/opt/rocm/lib/librccl.so.1"
    rccl = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)             # libpkv finds this copy by itself (no PKV_RCCL_LIB)

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    uid = UniqueId()
    rccl.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    torch.cuda.set_device(0)
    torch.zeros(1, device=DEV)                                    # HIP context
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    try:
        for B, Hl, k in ((1, 4, 120), (3, 2, 17)):
            idx = torch.randint(0, 32760, (B, Hl, k), dtype=torch.int32, device=DEV)
            out = torch.full((B, Hl, k), -1, dtype=torch.int32, device=DEV)
            ws = torch.empty(B * Hl * k * 4, dtype=torch.uint8, device=DEV)
            rc = N.lib.pkv_allgather_indices(comm, idx.data_ptr(), out.data_ptr(), B, Hl, k, ws.data_ptr(), ws.numel(),
                                             N.stream_ptr())
            assert rc == 0, (rc, N.lib.pkv_strerror(rc), N.lib.pkv_last_nccl_error())
            torch.cuda.synchronize()
            assert torch.equal(out, idx)
        assert N.lib.pkv_allgather_indices(None, idx.data_ptr(), out.data_ptr(), 1, 1, 1, None, 0, N.stream_ptr()) == -7
    finally:
        rccl.ncclCommDestroy(comm)
