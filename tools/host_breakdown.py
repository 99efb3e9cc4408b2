"""Where the host time of one update_kv goes (round 5): the prepared path (ops.PreparedCompress / PreparedAda) against the
general one, the foreign call alone, and its Python ingredients.  Tiny S for the issue cost (the GPU never limits), S = 32768
for the Ada-SnapKV call with its host sync."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyramidkv_amd as P
from pyramidkv_amd import ops

res = {}
dev = torch.device("cuda", 0)


def per_call(fn, n=2000, sync_every=100):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t = 0.0
    done = 0
    while done < n:
        t0 = time.perf_counter()
        for _ in range(sync_every):
            fn()
        t += time.perf_counter() - t0
        torch.cuda.synchronize()
        done += sync_every
    return round(t / n * 1e6, 2)


for S in (512, 4096):
    q, k, v = (torch.randn(1, 32, S, 128, device=dev).to(torch.bfloat16) for _ in range(3))
    cl = P.SnapKVCluster(window_size=8, max_capacity_prompt=128, kernel_size=7, pooling="maxpool")
    cl.update_kv(k, q, v, None, 1)
    assert cl._prep is not None
    pc = cl._prep[1]
    r = {"update_kv_prepared_us": per_call(lambda: cl.update_kv(k, q, v, None, 1)),
         "ops_compress_general_us": per_call(lambda: ops.compress(q, k, v, 8, 120, "maxpool", 7))}
    st = ops._raw_stream(0)
    ws = ops._workspace_and_stream(pc.nb, dev)[0]
    ko = torch.empty(pc.out_shape, dtype=pc.dtype, device=dev)
    vo = torch.empty_like(ko)
    args = (pc.dref, q.data_ptr(), k.data_ptr(), v.data_ptr(), ko.data_ptr(), vo.data_ptr(), None, ws.data_ptr(), ws.numel(), st)
    r["foreign_call_pkv_compress_us"] = per_call(lambda: pc.fn(*args))
    r["hit_us"] = per_call(lambda: pc.hit(q, k, v), sync_every=1000)
    r["two_torch_empty_us"] = per_call(lambda: (torch.empty(pc.out_shape, dtype=pc.dtype, device=dev), torch.empty(pc.out_shape, dtype=pc.dtype, device=dev)), sync_every=1000)
    r["raw_stream_us"] = per_call(lambda: ops._raw_stream(0), sync_every=1000)
    r["five_data_ptr_us"] = per_call(lambda: (q.data_ptr(), k.data_ptr(), v.data_ptr(), ko.data_ptr(), vo.data_ptr()), sync_every=1000)
    res["snapkv_S%d" % S] = r

# Ada-SnapKV, BASELINE config 5 shapes: where the wall time of a call goes
Hq, Hkv, S = 32, 8, 32768
q = torch.randn(1, Hq, S, 128, device=dev).to(torch.bfloat16)
k, v = (torch.randn(1, Hkv, S, 128, device=dev).to(torch.bfloat16) for _ in range(2))
cl = P.AdaKVCluster(window_size=8, kernel_size=7, pooling="maxpool", max_capacity_prompt=128, floor=0.2, normalize=True)
for _ in range(5):
    cl.update_kv(k, q, v)
torch.cuda.synchronize()
mirror = cl._mirror
waits = []
orig_wait = mirror.wait_words            # what the fast path of update_kv calls


def timed_wait(device):
    t0 = time.perf_counter()
    out = orig_wait(device)
    waits.append(time.perf_counter() - t0)
    return out


mirror.wait_words = timed_wait
n = 200
t0 = time.perf_counter()
for _ in range(n):
    cl.update_kv(k, q, v)
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / n * 1e6
res["adakv_S32768_budget128"] = {"update_kv_us": round(tot, 2), "of_which_waiting_for_the_capacities_us": round(sum(waits) / n * 1e6, 2),
                                 "host_work_us": round(tot - sum(waits) / n * 1e6, 2), "prepared": cl.ada.prepared is not None,
                                 "list_len": cl.ada.prepared.m_use if cl.ada.prepared is not None else None}
print(json.dumps(res, indent=1))
