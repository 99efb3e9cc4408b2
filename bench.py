#!/usr/bin/env python
"""bench.py - headline benchmark of the KV-compress hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

Workload (BASELINE.json metric: "prefill tokens/s + KV-compress ms at S=32k budget=128, Llama-3-8B"):
  PyramidKV, budget 128, window 8, maxpool-7 (the reference runners' knobs, run_longbench.py:221,236-237),
  Llama-3-8B attention shapes H=32, D=128, S=32768, bf16, synthetic N(0,1) Q/K/V resident in HBM.
  One STEP = the compress work of one model prefill: 32 update_kv calls, layer budgets 234..17
  (pyramidkv_utils.py:205-215), each on a [B,H,S,D] batch.  value = B*S*steps / time.
Multi-GPU: head-sharded (rank r owns H/N heads of B=N sequences: per-GPU bytes fixed => weak scaling)
with one RCCL all-gather of the selected indices per layer (pyramidkv_amd/dist.py).

Rank 0 prints ONE JSON line with the contract fields plus `roofline` (dominant kernel = the K scan),
`roofline_kernels` (every kernel, incl. the gather-compaction the north star targets) and
`cpu_baseline` (the oracle = the reference's eager CPU path, timed on this node's host cores).
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
NUM_LAYERS = 32


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--seq", type=int, default=32768)
    ap.add_argument("--budget", type=int, default=128)
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--batch", type=int, default=0, help="0 = one sequence per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--policy", default="pyramidkv", choices=["pyramidkv", "snapkv"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--cpu-layers", type=int, default=32, help="layer-calls per CPU-baseline pass (32 = the whole step)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU work spent on the cpu_baseline sample")
    return ap.parse_args()


def layer_budgets(P, policy, cap, w, S):
    ks = []
    for layer in range(NUM_LAYERS):
        if policy == "pyramidkv":
            cl = P.PyramidKVCluster(num_hidden_layers=NUM_LAYERS, layer_idx=layer, window_size=w,
                                    max_capacity_prompt=cap, kernel_size=7, pooling="maxpool")
            _, k = cl.layer_budget(S)
        else:
            k = cap - w
        ks.append(k)
    return ks


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    import pyramidkv_amd as P
    from pyramidkv_amd import _native as N, dist as pdist

    # PKV_BENCH_BACKEND=gloo (+ all ranks on one GPU) exists only to exercise the N>1 code path on a 1-GPU box
    backend = os.environ.get("PKV_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)       # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    e = 2
    S, H, D, w, cap = a.seq, a.heads, 128, 8, a.budget
    B = a.batch if a.batch > 0 else world
    h0, h1 = pdist.shard_heads(H, rank, world)
    Hl = h1 - h0
    ks = layer_budgets(P, a.policy, cap, w, S)

    # synthetic inputs, resident in HBM before the timed region.  NSETS distinct (Q,K,V) sets are cycled
    # over the layers so that a layer never finds its K in the 256 MB Infinity Cache from the previous call.
    NSETS = 4
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    sets = []
    for _ in range(NSETS):
        q = torch.randn(B, Hl, S, D, generator=gen, device=dev, dtype=torch.float32).to(dt)
        k = torch.randn(B, Hl, S, D, generator=gen, device=dev, dtype=torch.float32).to(dt)
        v = torch.randn(B, Hl, S, D, generator=gen, device=dev, dtype=torch.float32).to(dt)
        sets.append((q, k, v))

    def one_step():
        # the one exchange step of the path (indices, KBs) is issued asynchronously: layer i+1's update_kv does not
        # depend on layer i's gathered indices, so the latency-bound collective overlaps with it; at most two are
        # in flight and all are waited for (on the stream) before the step ends
        outs, pending = None, []
        for layer in range(NUM_LAYERS):
            q, k, v = sets[layer % NSETS]
            kc, vc, idx = P.ops.compress(q, k, v, w, ks[layer], "maxpool", 7, return_indices=True)
            if world > 1:
                pending.append(pdist.allgather_indices_async(idx))
                if len(pending) > 2:
                    idx = pending.pop(0).wait()
            outs = (kc, vc, idx)
        for h in pending:
            outs = (outs[0], outs[1], h.wait())
        return outs

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_step()
    barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    ms_per_step = el / a.steps * 1e3
    tokens_per_s = B * S * a.steps / el

    # ---- per-kernel device time over the same K steps (hipEvent pairs recorded inside libpkv) ----
    N.prof_enable(True)
    N.prof_read(reset=True)
    for _ in range(a.steps):
        one_step()
    torch.cuda.synchronize()
    prof = N.prof_read(reset=True)
    N.prof_enable(False)

    def rl(name, bytes_per_launch):
        ms, n = prof[name]
        if n == 0:
            return None
        avg_ms = ms / n
        ach = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        return {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "avg_us": round(avg_ms * 1e3, 2),
                "launches": int(n), "algorithmic_bytes": int(bytes_per_launch)}

    rows_mean = sum(k + w for k in ks) / NUM_LAYERS
    kernels = {
        # K read once + the w query rows (S*D*e per head)
        "logits": rl("logits", B * Hl * (S * D * e + w * D * e)),
        # logits [w][S] read once, pooled scores written
        "finalize": rl("finalize", B * Hl * (w * S * e + (S - w) * e)),
        "topk": rl("topk", B * Hl * ((S - w) * e + (sum(ks) / NUM_LAYERS) * 4)),
        # the north-star kernel: 2 tensors x (k+w) rows x D x e x (read+write), mean over the 32 layer budgets
        "gather": rl("gather", 4 * rows_mean * D * e * B * Hl),
    }
    kernels = {k_: v_ for k_, v_ in kernels.items() if v_}
    # HBM traffic per launch from rocprofv3 PMC passes of this same command (tools/pmc_summary.py writes the file)
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path) and a.seq == 32768 and a.budget == 128 and B == 1:
        pmc = json.load(open(pmc_path))
        for k_, v_ in kernels.items():
            if k_ in pmc.get("kernels", {}):
                v_["traffic"] = pmc["kernels"][k_].get("hbm_bytes_per_launch")

    out = {
        "metric": "prefill tokens/s through KV-compress (PyramidKV budget=%d, S=%d, Llama-3-8B shapes)" % (cap, S),
        "value": round(tokens_per_s, 1), "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 4), "kv_compress_ms_per_layer": round(ms_per_step / NUM_LAYERS, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": "%s budget=%d window=8 maxpool7, 32 layer-calls/step, [B=%d,H=%d,S=%d,D=128] %s"
                               % (a.policy, cap, B, H, S, a.dtype),
                   "global_batch": B, "seq_len": S, "heads_per_gpu": Hl,
                   "parallelism": "head-shard x%d + 1 all-gather(indices)/layer" % world if world > 1 else "single GPU"},
        "roofline": kernels.get("logits"),
        "roofline_kernels": kernels,
    }

    # extra (not `value`): the same workload when K/V really are repeat_kv output of 8 KV heads (Llama-3-8B /
    # Mistral-7B) and the caller opts into reading one head per GQA group (config.gqa_dedup)
    if world == 1 and not a.no_extras and Hl % 4 == 0:
        gsets = []
        for (q, k, v) in sets:
            k4 = k[:, ::4][:, :, None].expand(B, Hl // 4, 4, S, D).reshape(B, Hl, S, D).contiguous()
            v4 = v[:, ::4][:, :, None].expand(B, Hl // 4, 4, S, D).reshape(B, Hl, S, D).contiguous()
            gsets.append((q, k4, v4))

        def gstep():
            for layer in range(NUM_LAYERS):
                q, k, v = gsets[layer % NSETS]
                P.ops.compress(q, k[:, ::4], v[:, ::4], w, ks[layer], "maxpool", 7, kv_group=4)
        gstep()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            gstep()
        torch.cuda.synchronize()
        ge = time.perf_counter() - t0
        out["extras"] = {"gqa_dedup_tokens_per_s": round(B * S * a.steps / ge, 1),
                         "gqa_dedup_us_per_layer": round(ge / a.steps / NUM_LAYERS * 1e6, 2),
                         "note": "K/V = repeat_kv of 8 KV heads, kernels read 1 head per group (opt-in config.gqa_dedup); not the headline value"}
        del gsets

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(sets[0], ks, w, cap, a)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(qkv, ks, w, cap, a):
    """The reference's eager path (oracle restatement == pyramidkv_utils.py:197-283 executed by PyTorch CPU) on
    the host cores of this node, same tensors moved to the CPU.  Bounded sample: a few of the 32 layer-calls,
    thread count chosen by a one-call probe over {all cores, 64, 16} (the fastest is reported in `cores`)."""
    from oracle import pkv_oracle as O
    q, k, v = (t[:1].cpu() for t in qkv)        # one sequence, this rank's heads
    ncpu = os.cpu_count() or 1
    S = q.shape[2]
    n = max(1, min(a.cpu_layers, NUM_LAYERS))
    layers = [round(i * (NUM_LAYERS - 1) / max(1, n - 1)) for i in range(n)] if n > 1 else [0]

    def run(ls):
        t0 = time.perf_counter()
        for layer in ls:
            with contextlib.redirect_stdout(io.StringIO()):
                if a.policy == "pyramidkv":
                    O.pyramidkv_update_kv(k, q, v, w, cap, 7, "maxpool", NUM_LAYERS, layer, topk_mode="reference")
                else:
                    O.snapkv_update_kv(k, q, v, w, cap, 7, "maxpool", topk_mode="reference")
        return time.perf_counter() - t0

    best_t, best_n = None, None
    for nt in sorted({ncpu, min(ncpu, 64), min(ncpu, 16)}, reverse=True):
        torch.set_num_threads(nt)
        run([0])                         # warm-up at this thread count
        t = run([0])
        if best_t is None or t < best_t:
            best_t, best_n = t, nt
    torch.set_num_threads(best_n)
    # bounded sample: whole passes over the chosen layer-calls until ~cpu_seconds of CPU work; the fastest pass counts
    # (the host is shared with the GPU driver threads, single passes scatter by 2x)
    t, spent, passes = None, 0.0, 0
    while spent < a.cpu_seconds or passes < 2:
        tp = run(layers)
        spent += tp
        passes += 1
        t = tp if t is None or tp < t else t
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(S * len(layers) / NUM_LAYERS / t, 1), "unit": "tokens/s", "cores": best_n, "kind": "port",
            "sample": "%d of the 32 layer-calls per pass (layers %s) of the same workload ([1,%d,%d,128] %s), %d passes = %.1f s "
                      "of CPU work, fastest pass %.2f s; host has %d logical CPUs, thread count picked by a 1-call probe"
                      % (len(layers), layers if len(layers) < NUM_LAYERS else "0..31", q.shape[1], S, a.dtype, passes, spent, t, ncpu),
            "ms_per_layer": round(t / len(layers) * 1e3, 3), "cpu": model}


if __name__ == "__main__":
    main()
