"""Copy the judged artefacts of tools/r05_full_session.sh from gpurun_out/full (scratch) into profiles/r05/ and write
SUMMARY.md.  Usage: python tools/collect_r05.py"""
import csv, glob, hashlib, json, os, shutil, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out", "full5"), os.path.join(ROOT, "profiles", "r05")
os.makedirs(P, exist_ok=True)


def src_sha():
    sys.path.insert(0, ROOT)
    from bench import kernel_src_sha16
    return kernel_src_sha16()


def first_json_line(path):
    if not os.path.exists(path):
        return None
    for line in open(path):
        line = line.strip()
        if line.startswith("{"):
            try:
                return json.loads(line)
            except ValueError:
                pass
    try:
        return json.load(open(path))
    except ValueError:
        return None


for name in ("bench.json", "sweep.json", "bw_probe.json", "parity_report.json", "policy_bench.json", "ada_bench.json", "topk_trace.json",
             "pytest.txt", "bench_rccl_n1.json", "bench_rccl_n1.log", "bench_n2_gloo_selflaunch.json", "bench_n2_gloo_selflaunch.log",
             "bench_n2_nccl_one_gpu.log", "parity_sweep.json", "soak.txt", "bench_n8_gloo_one_gpu.json"):
    if os.path.exists(os.path.join(G, name)):
        shutil.copy(os.path.join(G, name), os.path.join(P, name))
    elif name == "parity_report.json" and os.path.exists(os.path.join(ROOT, "gpurun_out", name)):
        shutil.copy(os.path.join(ROOT, "gpurun_out", name), os.path.join(P, name))     # the suite ran in its own gpurun call
lines = ["# Profiles, round r05\n", "All numbers measured on one MI355X (gfx950) through `gpurun` by `tools/r05_full_session.sh`; raw files sit next to "
         "this summary.  Kernel sources: sha16 `%s`.  The rocprofv3 --stats files, the issue-port counters, the parity sweep "
         "and the H2O / merge records were written by the full session at sha16 `d5358b59e01886d7`; the one source edit since "
         "is host code (`pkv_coll.hip`: which RCCL library gets opened) - no kernel changed - and the closing session "
         "(`tools/archive/r05_final_check3.sh`) re-ran the HBM traffic passes, the driver's bench command, the suite and smoke "
         "on the final library (`pmc_traffic.json`, `bench.json`, `pytest.txt`, `smoke.log`).\n" % src_sha()]


def stats_table(pattern, title, top=8):
    st = sorted(glob.glob(os.path.join(G, pattern, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if not st:
        return
    rows = list(csv.DictReader(open(st[-1])))
    dst = pattern + "_kernel_stats.csv"
    with open(os.path.join(P, dst), "w", newline="") as f:
        wr = csv.writer(f)
        wr.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            wr.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
    lines.extend(["## " + title + "  (`%s`)\n" % dst, "| kernel | calls | avg us | min us | max us | % of GPU time |", "|---|---|---|---|---|---|"])
    for r in rows[:top]:
        lines.append("| %s | %s | %.2f | %.2f | %.2f | %s |" % (r["Name"].split("(")[0][-56:], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                            float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
    lines.append("")


stats_table("prof_headline", "rocprofv3 --kernel-trace --stats -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras (headline workload only)", 6)
stats_table("prof", "rocprofv3 --kernel-trace --stats -- python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline (all legs of the bench line mixed: B = 1 and B = 8, budgets 128 and 2048, un-expanded K)", 10)
# the gather kernel of the full run split by launch grid: gather_kernel<8> with 16 row blocks x 32 heads = 512 workgroups is budget 2048 at
# B = 1, 4096 workgroups is B = 8
tr = sorted(glob.glob(os.path.join(G, "prof", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
if tr:
    by = defaultdict(list)
    for r in csv.DictReader(open(tr[-1])):
        n = r["Kernel_Name"]
        if "pkv::gather_kernel" in n:
            by[(n.split("pkv::")[1].split("(")[0], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    rows = [{"kernel": k[0], "workgroups": k[1], "launches": len(v), "avg_us": round(sum(v) / len(v), 2), "min_us": round(min(v), 2)} for k, v in sorted(by.items())]
    json.dump(rows, open(os.path.join(P, "rocprofv3_gather_by_grid.json"), "w"), indent=1)
    lines += ["### gather_kernel of that run by launch grid (`rocprofv3_gather_by_grid.json`; 512 workgroups of gather_kernel<8> = budget 2048 at B = 1, 4096 = B = 8)\n",
              "| kernel | workgroups | launches | avg us | min us |", "|---|---|---|---|---|"]
    for r in rows:
        lines.append("| %s | %d | %d | %.2f | %.2f |" % (r["kernel"], r["workgroups"], r["launches"], r["avg_us"], r["min_us"]))
    lines.append("")
pm = {}
pj = os.path.join(G, "pmc_traffic.json")
if os.path.exists(pj):
    t = json.load(open(pj))
    t["kernel_src_sha16"] = src_sha()
    json.dump(t, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
    pm = t["kernels"]
b = first_json_line(os.path.join(G, "bench.json"))
if b:
    lines += ["## bench.py --gpus 1 --steps 20 --warmup 5 (the driver's command)\n",
              "`%s`: **%.4g tokens/s**, %.1f us per update_kv (`%s`); whole-call effective %.0f GB/s = %.3f of 8 TB/s\n"
              % (b["metric"], b["value"], b["kv_compress_ms_per_layer"] * 1e3, b["config"]["workload"], b["call_effective"]["GBps"], b["call_effective"]["frac_of_8TBps"]),
              "| kernel | avg us (events on the dispatch) | algorithmic MB | achieved GB/s | frac of 8 TB/s | PMC HBM MB/launch |", "|---|---|---|---|---|---|"]
    for k, v in b["roofline_kernels"].items():
        t = pm.get(k, {}).get("hbm_bytes_per_launch")
        lines.append("| %s | %.2f | %.2f | %.0f | %.3f | %s |" % (k, v["avg_us"], v["algorithmic_bytes"] / 1e6, v["achieved"], v["frac"],
                                                                 "%.2f" % (t / 1e6) if t else "-"))
    lines += ["", "### Grid: one SnapKV update_kv call, B x budget (S = 32768, H = 32, bf16)\n",
              "| B | budget | update_kv us | call eff. of 8 TB/s | logits us | finalize us | topk us | gather us | gather frac |", "|---|---|---|---|---|---|---|---|---|"]
    for r in b.get("grid", []):
        lines.append("| %d | %d | %.1f | %.3f | %.1f | %.1f | %.1f | %.1f | %.3f |" % (r["B"], r["budget"], r["update_kv_us"], r["call_effective_frac_of_8TBps"],
                                                                                     r["logits"]["us"], r["finalize"]["us"], r["topk"]["us"], r["gather"]["us"], r["gather"]["frac"]))
    ge, ex, c = b.get("gpu_eager_baseline"), b.get("extras"), b.get("cpu_baseline")
    if ge:
        lines += ["", "Same-chip comparator (reference op sequence, PyTorch-ROCm eager): budget 128 %.1f us, budget 2048 %.1f us per update_kv (%s)\n"
                  % (ge["snapkv_budget128"]["update_kv_us"], ge["snapkv_budget2048"]["update_kv_us"], ge["kind"])]
    if ex:
        lines += ["K/V handed over before repeat_kv (8 KV heads): %.4g tokens/s, %.1f us per update_kv\n" % (ex["unexpanded_gqa_tokens_per_s"], ex["unexpanded_gqa_us_per_layer"])]
    if c:
        lines += ["CPU baseline (`%s`, %d threads, %s): %.4g tokens/s, %.1f ms per update_kv; %s; sample: %s\n"
                  % (c["kind"], c["cores"], c.get("cpu", ""), c["value"], c["ms_per_layer"], c.get("port_checked_against", ""), c["sample"])]
if b and b.get("parity"):
    pr = b["parity"]
    lines += ["### Parity of the timed step (bench line field `parity`: layers 0 and 31 of the step function that was timed, vs the CPU oracle)\n",
              "heads with the oracle's index SET %.3f, index SEQUENCE %.3f, K/V bits %.3f; largest order inversion %d ulp (%s)\n"
              % (pr["heads_identical_set"], pr["heads_identical_sequence"], pr["kv_bit_identical_heads"], pr["max_order_inversion_ulp"], pr["shape"])]
if b and b.get("sweep"):
    lines += ["### BASELINE.json's synthetic sweep (one SnapKV budget-128 update_kv per point)\n", "| B | S | update_kv us | tokens/s | call effective of 8 TB/s |", "|---|---|---|---|---|"]
    for r in b["sweep"]:
        lines.append("| %d | %d | %.1f | %.3g | %.3f |" % (r["B"], r["S"], r["update_kv_us"], r["tokens_per_s"], r["call_effective_frac_of_8TBps"]))
    lines.append("")
ps = first_json_line(os.path.join(G, "parity_sweep.json"))
if ps:
    sm = ps["summary"]
    lines += ["## Parity over the whole sweep (tools/parity_sweep.py; `parity_sweep.json`): B x S x budget x dtype, H = 32, HIP path vs the CPU oracle\n",
              "%d points, %d heads: identical SET %.5f, identical SEQUENCE %.5f (budget 128: %.5f, budget 2048: %.5f), K/V bits %.5f; largest inversion of the oracle's scores in the kernel's order: %d ulp\n"
              % (sm["points"], sm["heads"], sm["set_rate"], sm["sequence_rate"], sm["by_budget"]["128"]["sequence_rate"], sm["by_budget"]["2048"]["sequence_rate"],
                 sm["kv_bit_identical_rate"], sm["max_order_inversion_ulp"]),
              "| dtype | budget | B | S=4096 | S=8192 | S=16384 | S=32768 |   (heads with the oracle's sequence / heads)", "|---|---|---|---|---|---|---|"]
    tab = {}
    for r in ps["rows"]:
        tab[(r["dtype"], r["budget"], r["B"], r["S"])] = "%d/%d%s" % (r["heads_identical_sequence"], r["heads"], "" if r["heads_identical_set"] == r["heads"] else " (set %d)" % r["heads_identical_set"])
    for dtn in ("bfloat16", "float16"):
        for cap in (128, 2048):
            for B in (1, 2, 4, 8):
                lines.append("| %s | %d | %d | %s |" % (dtn, cap, B, " | ".join(tab.get((dtn, cap, B, S), "-") for S in (4096, 8192, 16384, 32768))))
    lines.append("")
for name, title in (("bench_rccl_n1.json", "RCCL, nranks = 1, one all-gather per prefill"),
                    ("bench_n2_gloo_selflaunch.json", "`PKV_BENCH_BACKEND=gloo python bench.py --gpus 2` from a plain shell: bench.py starts its own two ranks (both share the one GPU - a code-path run, NOT a scaling number)")):
    j = first_json_line(os.path.join(G, name))
    if j:
        lines += ["## %s\n" % title, "%.4g tokens/s, %.1f us per update_kv, parallelism `%s`, backend `%s`, rccl_nranks %s%s\n"
                  % (j["value"], j["kv_compress_ms_per_layer"] * 1e3, j["config"]["parallelism"], j["config"].get("collective_backend"), j.get("rccl_nranks"),
                     (", legs: " + json.dumps(j["scaling_legs"])) if "scaling_legs" in j else "")]
s = first_json_line(os.path.join(G, "sweep.json"))
if s:
    lines += ["## Per-kernel sweep (SnapKV, H=32, D=128, bf16, w=8, maxpool-7; us and GB/s of algorithmic bytes)\n",
              "| config | update_kv us | tokens/s | logits | finalize | topk | gather | gather frac of 8 TB/s |", "|---|---|---|---|---|---|---|---|"]
    for k, v in s.items():
        lines.append("| %s | %.1f | %.3g | %.1f us / %.0f | %.1f us | %.1f us | %.1f us / %.0f | %.3f |" % (
            k, v["update_kv_us"], v["tokens_per_s"], v["logits"]["us"], v["logits"]["GBps"], v["finalize"]["us"], v["topk"]["us"],
            v["gather"]["us"], v["gather"]["GBps"], v["gather"]["frac_of_8TBps"]))
    lines.append("")
pb = first_json_line(os.path.join(G, "policy_bench.json"))
ab = first_json_line(os.path.join(G, "ada_bench.json"))
if pb or ab:
    lines += ["## Other policies (B = 1, H = 32, bf16)\n", "| case | update_kv ms | kernels us |", "|---|---|---|"]
    for src in (pb, ab):
        for k, v in (src or {}).items():
            lines.append("| %s | %s | %s |" % (k, v.get("update_kv_ms"), json.dumps(v.get("kernels_us"))))
    lines.append("")
# issue-port counters of the headline kernels
acc_i = defaultdict(lambda: defaultdict(list))
for d in ("pmc_issue", "pmc_issue2"):
    for f in sorted(glob.glob(os.path.join(G, d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(f)):
            if "pkv::" in r["Kernel_Name"]:
                acc_i[r["Kernel_Name"].split("pkv::")[1].split("<")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
if acc_i:
    iss = {k: {c: sum(x) / len(x) for c, x in v.items()} for k, v in acc_i.items()}
    json.dump(iss, open(os.path.join(P, "pmc_issue.json"), "w"), indent=1)
    cols = sorted({c for v in iss.values() for c in v})
    lines += ["## Issue-port counters of the headline kernels (rocprofv3 --pmc, two passes, per launch, summed over the chip; `pmc_issue.json`)\n",
              "| kernel | " + " | ".join(cols) + " |", "|---|" + "---|" * len(cols)]
    for k, v in iss.items():
        lines.append("| %s | " % k + " | ".join("%.4g" % v.get(c, float("nan")) for c in cols) + " |")
    lines += ["", "SQ_INSTS_VALU / (waves of the launch) = vector instructions per wave: finalize_kernel launches 1024 workgroups x 4 waves, "
              "topk_kernel 32 x 16, gather_kernel<2,16> ~256 x 4, logits2_kernel 8192 x 4 at the headline shape.", ""]
stats_table("prof_ada", "Ada-SnapKV / HeadKV (tools/ada_bench.py), rocprofv3 --kernel-trace --stats", 12)
stats_table("prof_h2o", "H2O at S = 32768 (tools/h2o_only.py), rocprofv3 --kernel-trace --stats", 4)
acc = defaultdict(lambda: defaultdict(list))
for d in ("pmc_h2o_a", "pmc_h2o_b", "pmc_h2o_c"):
    for f in sorted(glob.glob(os.path.join(G, d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]:   # newest run only: gpurun merges, never cleans
        for r in csv.DictReader(open(f)):
            if "pkv::" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("pkv::")[1].split("<")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
if acc:
    h2o = {k: {c: sum(x) / len(x) for c, x in v.items()} for k, v in acc.items()}
    json.dump(h2o, open(os.path.join(P, "pmc_h2o.json"), "w"), indent=1)
    lines += ["## H2O issue-port counters (rocprofv3 --pmc, two SQ passes, per launch; `pmc_h2o.json`)\n", "| kernel | " + " | ".join(sorted(next(iter(h2o.values())))) + " |",
              "|---|" + "---|" * len(next(iter(h2o.values())))]
    for k, v in h2o.items():
        lines.append("| %s | " % k + " | ".join("%.4g" % v[c] for c in sorted(v)) + " |")
    lines.append("")
    bf = os.path.join(ROOT, "profiles", "r03", "pmc_h2o.json")
    if os.path.exists(bf):
        b4 = json.load(open(bf))
        lines += ["Round 3's H2O kernels (`../r03/pmc_h2o.json`) -> now (`h2o/h2o_account.md` reads these numbers):\n", "| kernel | SQ_INSTS_VALU | SQ_ACTIVE_INST_VALU | SQ_BUSY_CYCLES |", "|---|---|---|---|"]
        for k in h2o:
            if k in b4:
                lines.append("| %s | " % k + " | ".join("%.4g -> %.4g" % (b4[k][c], h2o[k][c]) for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES")) + " |")
        lines.append("")
ex5 = (b or {}).get("extras") or {}
if ex5.get("config2_pyramidkv_8k"):
    c2 = ex5["config2_pyramidkv_8k"]
    lines += ["## BASELINE config 2 on the driver's line (`extras.config2_pyramidkv_8k`): PyramidKV budget 128, S = 8192, all 32 layer budgets\n",
              "%.1f us per update_kv (host issue %.1f us, kernels %s = %.1f us), K scan %.3f of 8 TB/s, call effective %.3f; parity set / sequence / K,V bits %s\n"
              % (c2["update_kv_us"], c2["host_us"], json.dumps(c2["kernels_us"]), c2["kernels_sum_us"], (c2.get("roofline") or {}).get("frac", float("nan")),
                 c2["call_effective_frac_of_8TBps"], "%.3f / %.3f / %.3f" % tuple(c2["parity"][k_] for k_ in ("heads_identical_set", "heads_identical_sequence", "kv_bit_identical_heads")) if c2.get("parity") else "-")]
if ex5.get("config5_adakv_gqa_32k"):
    c5 = ex5["config5_adakv_gqa_32k"]
    lines += ["## BASELINE config 5 on the driver's line (`extras.config5_adakv_gqa_32k`): %s\n" % c5["workload"],
              "| budget | update_kv us | kernels us per call | kernels sum | host / sync remainder | call over its algorithmic bytes, of 8 TB/s | parity (budgets / metadata / K,V heads) |", "|---|---|---|---|---|---|---|"]
    for cap in (128, 2048):
        r5 = c5.get("budget%d" % cap)
        if r5:
            pr5 = r5.get("parity") or {}
            lines.append("| %d | %.1f | %s | %.1f | %.1f | %.3f | %s / %s / %s |" % (cap, r5["update_kv_us"], json.dumps(r5["kernels_us_per_call"]), r5["kernels_sum_us"],
                         r5["host_sync_remainder_us"], r5["roofline"]["frac"], pr5.get("head_budgets_identical"), pr5.get("metadata_identical"), pr5.get("kv_bit_identical_heads")))
    lines.append("")
if ex5.get("merge"):
    lines += ["## LOOK-M merge on the driver's line (`extras.merge`)\n", "| case | plain update_kv us | merge update_kv us | merge step us | merge step of 8 TB/s |", "|---|---|---|---|---|"]
    for k_, v_ in ex5["merge"].items():
        lines.append("| %s | %.1f | %.1f | %.1f | %.3f |" % (k_, v_["update_kv_plain_us"], v_["update_kv_merge_us"], v_["merge_only_us"], v_["roofline"]["frac"]))
    lines.append("")
drv = [first_json_line(os.path.join(ROOT, f)) for f in ("BENCH_r03.json", "BENCH_r04.json")]
if b:
    keys = ["logits", "finalize", "topk", "gather", "gather_cap2048_B1", "gather_cap2048_B8"]
    lines += ["## Driver-run vs builder-run, the six roofline rows (frac of 8 TB/s; avg us)\n",
              "| row | driver r03 | driver r04 | builder r05 (this session) |", "|---|---|---|---|"]
    def driver_line(j):
        """the JSON line the driver captured from its own bench.py run (BENCH_rNN.json: run.stdout_tail)"""
        for line in ((j.get("run") or {}).get("stdout_tail") or "").splitlines():
            if line.startswith("{"):
                try:
                    return json.loads(line)
                except ValueError:
                    pass
        return {}
    def cell(j, k_):
        try:
            v = (driver_line(j) if "run" in j else j)["roofline_kernels"][k_]
            return "%.3f (%.1f us)" % (v["frac"], v["avg_us"])
        except Exception:
            return "-"
    for k_ in keys:
        lines.append("| %s | %s | %s | %s |" % (k_, cell(drv[0] or {}, k_), cell(drv[1] or {}, k_), cell(b, k_)))
    lines += ["", "(the budget-2048 gather is the north-star row: target >= 0.60 at B = 1; box-to-box spread of the builder's own sessions this round: see `ab/`)", ""]
bw = os.path.join(G, "bw_probe.json")
if os.path.exists(bw):
    lines += ["## Achievable HBM bandwidth on this box (tools/bw_probe.hip, 1 GiB)\n", "```", open(bw).read().strip(), "```", ""]
tt = first_json_line(os.path.join(G, "topk_trace.json"))
if tt:
    lines += ["## top-k phase stamps inside update_kv (shader clock, row 0; debug build)\n", "| budget k | candidates | total cycles | stamps |", "|---|---|---|---|"]
    for k, v in tt.items():
        lines.append("| %s | %s | %s | %s |" % (k, v["C"], v["total"], v["stamps_rel"]))
    lines.append("")
mb, ms_ = os.path.join(G, "merge_bench.json"), os.path.join(G, "merge_kernel_split.txt")
if os.path.exists(mb) and os.path.getsize(mb):
    shutil.copy(mb, os.path.join(P, "merge_bench.json"))
    m = json.load(open(mb))
    old = os.path.join(ROOT, "profiles", "r03", "merge_bench.json")
    o = json.load(open(old)) if os.path.exists(old) else {}
    lines += ["## LOOK-M merge (tools/merge_bench.py, H = 32, D = 128, bf16; ms per call)\n",
              "| S, budget | update_kv plain | update_kv merge | merge alone | merge alone, round 3 | reference ops, eager on this GPU |", "|---|---|---|---|---|---|"]
    for k_, v in m.items():
        lines.append("| %s | %.4f | %.4f | %.4f | %s | %.3f |" % (k_, v["update_kv_plain_ms"], v["update_kv_merge_ms"], v["merge_only_ms"],
                                                              ("%.4f" % o[k_]["merge_only_ms"]) if k_ in o else "-", v["reference_ops_eager_merge_ms"]))
    lines.append("")
    if os.path.exists(ms_):
        shutil.copy(ms_, os.path.join(P, "merge_kernel_split.txt"))
        lines += ["Per-kernel split (rocprofv3 --kernel-trace --stats -- python tools/merge_only.py S budget; `merge_kernel_split.txt`):\n", "```", open(ms_).read().strip(), "```", ""]
abdir = os.path.join(P, "ab")
if os.path.isdir(abdir):
    lines += ["## Same-session A/B files of this round (`ab/`)\n"] + ["* `ab/%s`" % f_ for f_ in sorted(os.listdir(abdir))] + [""]
extra = [("prof_ada_bench_kernel_stats.csv", "rocprofv3 --kernel-trace --stats -- python tools/ada_bench.py (Ada-SnapKV / HeadKV, S = 8192 and 32768, budgets 128 and 2048)"),
         ("prof_h2o_S8192_kernel_stats.csv", "rocprofv3 --kernel-trace --stats -- python tools/h2o_only.py 8192"),
         ("bench_default_flags.json", "python bench.py (no flags: the driver's N = 1 form) on the final kernel sources"),
         ("parity_fuzz.txt", "tools/parity_fuzz.py over three new seeds + the replayed outlier (tools/fuzz_explain.py)"),
         ("pytest_after_full_session.txt", "the one GPU test added after the full session (host-side change), run in its own call")]
have = [(f_, d_) for f_, d_ in extra if os.path.exists(os.path.join(P, f_))]
if have:
    lines += ["## Further files of this round (own short sessions)\n"] + ["* `%s` - %s" % fd for fd in have] + [""]
sm = os.path.join(G, "smoke.log")
if os.path.exists(sm):
    shutil.copy(sm, os.path.join(P, "smoke.log"))
py = os.path.join(G, "pytest.txt")
if os.path.exists(py):
    lines += ["## GPU test suite\n", "```", "".join(open(py).readlines()[-12:]).strip(), "```", ""]
open(os.path.join(P, "SUMMARY.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
