"""Copy the judged artefacts of tools/r02_full_session.sh from gpurun_out/full (scratch) into profiles/r02/ and write
SUMMARY.md.  Usage: python tools/collect_r02.py"""
import csv, glob, hashlib, json, os, shutil, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out", "full"), os.path.join(ROOT, "profiles", "r02")
os.makedirs(P, exist_ok=True)


def src_sha():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "pyramidkv_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


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
             "pytest.txt", "bench_rccl_n1.json", "bench_rccl_n1.log", "bench_rccl_n1_perlayer.json", "bench_n2_gloo.json"):
    if os.path.exists(os.path.join(G, name)):
        shutil.copy(os.path.join(G, name), os.path.join(P, name))
    elif name == "parity_report.json" and os.path.exists(os.path.join(ROOT, "gpurun_out", name)):
        shutil.copy(os.path.join(ROOT, "gpurun_out", name), os.path.join(P, name))     # the suite ran in its own gpurun call
lines = ["# Profiles, round r02\n", "All numbers measured on one MI355X (gfx950) through `gpurun` by `tools/r02_full_session.sh`; raw files sit next to "
         "this summary.  Kernel sources: sha16 `%s`.\n" % src_sha()]


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
# gather shape A/B of session 1 (rows per lane x XCD-aligned block map)
shape = {}
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "s1", "sweep_rpt*_xcd*.json"))):
    j = first_json_line(f)
    if j:
        shape[os.path.basename(f)[6:-5]] = {k: {"gather_us": v["gather"]["us"], "frac_of_8TBps": v["gather"]["frac_of_8TBps"]} for k, v in j.items()}
if shape:
    json.dump(shape, open(os.path.join(P, "gather_shape.json"), "w"), indent=1)
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
for name, title in (("bench_rccl_n1.json", "RCCL, nranks = 1, one all-gather per prefill"), ("bench_rccl_n1_perlayer.json", "RCCL, nranks = 1, one all-gather per layer"),
                    ("bench_n2_gloo.json", "N = 2 code path on one GPU (gloo; both ranks share the device - NOT a scaling number)")):
    j = first_json_line(os.path.join(G, name))
    if j:
        lines += ["## %s\n" % title, "%.4g tokens/s, %.1f us per update_kv, parallelism `%s`, backend `%s`%s\n"
                  % (j["value"], j["kv_compress_ms_per_layer"] * 1e3, j["config"]["parallelism"], j["config"].get("collective_backend"),
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
    bf = os.path.join(P, "pmc_h2o_before.json")
    if os.path.exists(bf):
        b4 = json.load(open(bf))
        lines += ["Before this round's H2O rework (`pmc_h2o_before.json`, same command on commit d56ac47) -> now:\n", "| kernel | SQ_INSTS_VALU | SQ_ACTIVE_INST_VALU | SQ_BUSY_CYCLES |", "|---|---|---|---|"]
        for k in h2o:
            if k in b4:
                lines.append("| %s | " % k + " | ".join("%.4g -> %.4g" % (b4[k][c], h2o[k][c]) for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES")) + " |")
        lines.append("")
hab = os.path.join(G, "h2o_ab.txt")
if not os.path.exists(hab):
    hab = os.path.join(ROOT, "gpurun_out", "h2o_ab.txt")       # tools/r02_h2o_ab.sh run on its own
if os.path.exists(hab):
    shutil.copy(hab, os.path.join(P, "h2o_ab.txt"))
    lines += ["## H2O kernels before / after, same box (tools/r02_h2o_ab.sh; us per launch, S = 32768, H = 32)\n", "```", open(hab).read().strip(), "```", ""]
bw = os.path.join(G, "bw_probe.json")
if os.path.exists(bw):
    lines += ["## Achievable HBM bandwidth on this box (tools/bw_probe.hip, 1 GiB)\n", "```", open(bw).read().strip(), "```", ""]
tt = first_json_line(os.path.join(G, "topk_trace.json"))
if tt:
    lines += ["## top-k phase stamps inside update_kv (shader clock, row 0; debug build)\n", "| budget k | candidates | total cycles | stamps |", "|---|---|---|---|"]
    for k, v in tt.items():
        lines.append("| %s | %s | %s | %s |" % (k, v["C"], v["total"], v["stamps_rel"]))
    lines.append("")
sm = os.path.join(G, "smoke.log")
if os.path.exists(sm):
    shutil.copy(sm, os.path.join(P, "smoke.log"))
py = os.path.join(G, "pytest.txt")
if os.path.exists(py):
    lines += ["## GPU test suite\n", "```", "".join(open(py).readlines()[-12:]).strip(), "```", ""]
open(os.path.join(P, "SUMMARY.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
