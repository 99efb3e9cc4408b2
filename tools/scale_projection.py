#!/usr/bin/env python
"""PROJECTED 1/2/4/8-GPU line of BASELINE config 4 from ONE GPU (round-5 review item 8).  No scaling curve has been measured:
`gpurun` exposes one MI355X.  What CAN be measured on one GPU is every term of the multi-GPU step except the xGMI hop:

  shard_step_ms   the step of ONE rank of an N-rank run, on this GPU: 32 update_kv calls on that rank's shard - 32/N query heads
                  and their 8/N KV heads UN-EXPANDED ([1,32/N,S,128] q, [1,8/N,S,128] k/v, kv_group 4), PyramidKV budget 128 -
                  exactly what bench.py --gpus N times per rank (all ranks do the same work on different heads);
  host_issue_ms   the host time to issue those 32 calls (one process per GPU: not shared between ranks);
  allgather_us    one RCCL all-gather of the whole prefill's indices with nranks = 1 (bench.py under torch.distributed.run,
                  `allgather_us`): the collective's launch + kernel floor WITHOUT any link traversal.

  projected step = max(shard_step_ms, host_issue_ms) + allgather_us;  projected tokens/s = S / step.

The projection is an UPPER bound of the real line: it has no xGMI latency (a latency-bound 3.7 KB-per-rank ring over 7 links
adds microseconds per hop), no rank skew at the barrier, and every rank's K scan runs against an otherwise idle HBM.
Writes gpurun_out/scale_projection.json; every number in it is labelled measured-on-one-GPU or projected."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pyramidkv_amd as P

S, H, HKV, W, NL = 32768, 32, 8, 8, 32
dev = torch.device("cuda", 0)


def shard_step(N):
    Hl, Hk = H // N, max(1, HKV // N)
    g = torch.Generator(device=dev).manual_seed(1234)
    sets = [(torch.randn(1, Hl, S, 128, generator=g, device=dev).to(torch.bfloat16),
             torch.randn(1, Hk, S, 128, generator=g, device=dev).to(torch.bfloat16),
             torch.randn(1, Hk, S, 128, generator=g, device=dev).to(torch.bfloat16)) for _ in range(4)]
    cls = [P.PyramidKVCluster(num_hidden_layers=NL, layer_idx=i, window_size=W, max_capacity_prompt=128, kernel_size=7, pooling="maxpool")
           for i in range(NL)]

    def step():
        for i in range(NL):
            q, k, v = sets[i % 4]
            cls[i].update_kv(k, q, v, None, Hl // Hk)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    best = None
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 10
        best = t if best is None or t < best else best
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return best * 1e3, host * 1e3, Hl, Hk


def allgather_nranks1():
    """bench.py as rank 0 of a 1-rank RCCL group: its `allgather_us`."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--no-extras", "--no-parity"], capture_output=True, text=True, env=env, timeout=600)
    for line in r.stdout.splitlines():
        if line.startswith("{"):
            return json.loads(line).get("allgather_us")
    return None


def main():
    ag = allgather_nranks1()
    rows = []
    for N in (1, 2, 4, 8):
        ms, host, Hl, Hk = shard_step(N)
        step = max(ms, host) + (ag or 0.0) / 1e3 * (1 if N > 1 else 0)
        rows.append({"n_gpus": N, "label": "measured on one GPU" if N == 1 else "PROJECTED from one-GPU shard measurements (no xGMI term)",
                     "heads_per_gpu": Hl, "kv_heads_per_gpu": Hk,
                     "shard_step_ms_measured_on_one_gpu": round(ms, 4), "host_issue_ms_measured": round(host, 4),
                     "allgather_us_nranks1_measured": ag, "projected_ms_per_step": round(step, 4),
                     "projected_tokens_per_s": round(S / step * 1e3, 0), "bound": "host issue" if host > ms else "device"})
        torch.cuda.empty_cache()
    base = rows[0]["projected_tokens_per_s"]
    for r in rows:
        r["projected_speedup_vs_1gpu"] = round(r["projected_tokens_per_s"] / base, 3)
    out = {"what": "BASELINE config 4 (PyramidKV budget 128, S = 32768, one sequence head-sharded over N GPUs, strong scaling): PROJECTED line",
           "no_scaling_curve_has_been_measured": True, "device": torch.cuda.get_device_name(0), "rows": rows, "method": __doc__.split("\n\n")[1]}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "scale_projection.json"), "w"), indent=1)
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
