"""Copy the judged artefacts of the last full GPU session from gpurun_out/ (scratch) into profiles/<round>/ and
write a human-readable SUMMARY.md.  Usage: python tools/collect_profiles.py r01"""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles", rnd)
os.makedirs(P, exist_ok=True)
for name in ("bench.json", "sweep.json", "pmc_traffic.json", "bw_probe.json", "parity_report.json", "probe.json", "wg_trace.json",
             "policy_bench.json", "topk_trace.json"):
    if os.path.exists(os.path.join(G, name)):
        shutil.copy(os.path.join(G, name), os.path.join(P, name))
stats = sorted(glob.glob(os.path.join(G, "prof", "*", "*kernel_stats.csv")), key=os.path.getmtime)
lines = ["# Profiles, round %s\n" % rnd, "All numbers measured on one MI355X (gfx950) through `gpurun`; raw files sit next to this summary.\n"]
if stats:
    rows = list(csv.DictReader(open(stats[-1])))
    with open(os.path.join(P, "rocprofv3_kernel_stats.csv"), "w", newline="") as f:
        wr = csv.writer(f)
        wr.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            wr.writerow([r["Name"][:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
    lines += ["## rocprofv3 --kernel-trace --stats  (`python bench.py --steps 2 --warmup 1 --no-cpu-baseline`)\n",
              "| kernel | calls | avg us | min us | max us | % of GPU time |", "|---|---|---|---|---|---|"]
    for r in rows[:6]:
        lines.append("| %s | %s | %.2f | %.2f | %.2f | %s |" % (r["Name"].split("(")[0][-48:], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                            float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
    lines.append("")
bj = os.path.join(G, "bench.json")
if os.path.exists(bj):
    b = json.load(open(bj))
    lines += ["## bench.py (N=1)\n", "`%s`: **%.3g tokens/s**, %.1f us per layer-call (`%s`)\n" % (b["metric"], b["value"], b["kv_compress_ms_per_layer"] * 1e3, b["config"]["workload"]),
              "| kernel | avg us (hipEvent in libpkv) | algorithmic MB | achieved GB/s | frac of 8 TB/s | PMC HBM MB/launch |", "|---|---|---|---|---|---|"]
    pm = json.load(open(os.path.join(G, "pmc_traffic.json")))["kernels"] if os.path.exists(os.path.join(G, "pmc_traffic.json")) else {}
    for k, v in b["roofline_kernels"].items():
        t = pm.get(k, {}).get("hbm_bytes_per_launch")
        lines.append("| %s | %.2f | %.2f | %.0f | %.3f | %s |" % (k, v["avg_us"], v["algorithmic_bytes"] / 1e6, v["achieved"], v["frac"],
                                                                 "%.2f" % (t / 1e6) if t else "-"))
    c = b.get("cpu_baseline")
    if c:
        lines += ["", "CPU baseline (`%s`, %d threads, %s): %.4g tokens/s, %.1f ms per layer-call; sample: %s\n" % (c["kind"], c["cores"], c.get("cpu", ""), c["value"], c["ms_per_layer"], c["sample"])]
sj = os.path.join(G, "sweep.json")
if os.path.exists(sj):
    s = json.load(open(sj))
    lines += ["## Per-kernel sweep (SnapKV, H=32, D=128, bf16, w=8, maxpool-7; us and GB/s of algorithmic bytes)\n",
              "| config | update_kv us | tokens/s | logits | finalize | topk | gather | gather frac of 8 TB/s |", "|---|---|---|---|---|---|---|---|"]
    for k, v in s.items():
        lines.append("| %s | %.1f | %.3g | %.1f us / %.0f | %.1f us | %.1f us | %.1f us / %.0f | %.3f |" % (
            k, v["update_kv_us"], v["tokens_per_s"], v["logits"]["us"], v["logits"]["GBps"], v["finalize"]["us"], v["topk"]["us"],
            v["gather"]["us"], v["gather"]["GBps"], v["gather"]["frac_of_8TBps"]))
    lines.append("")
bw = os.path.join(G, "bw_probe.json")
if os.path.exists(bw):
    lines += ["## Achievable HBM bandwidth on this box (tools/bw_probe.hip, 1 GiB)\n", "```", open(bw).read().strip(), "```", ""]
open(os.path.join(P, "SUMMARY.md"), "w").write("\n".join(lines) + "\n")
if os.path.exists(os.path.join(G, "pmc_traffic.json")):
    shutil.copy(os.path.join(G, "pmc_traffic.json"), os.path.join(ROOT, "profiles", "pmc_traffic.json"))
print("\n".join(lines))
