"""Kernel timeline of the bench from a rocprofv3 --kernel-trace CSV: durations of the pkv kernels and the gaps
between consecutive pkv kernels (end -> next start), per kernel pair.  Usage: timeline.py <dir with *_kernel_trace.csv>"""
import csv, glob, json, os, sys
import numpy as np

def short(n):
    for k in ("logits2_kernel", "logits_kernel", "finalize_kernel", "topk_kernel", "gather_kernel", "h2o_stats", "h2o_colsum",
              "sort_rows", "ada_"):
        if k in n:
            return k
    return None

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            s = short(r["Kernel_Name"])
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), s))
rows.sort()
dur, gap = {}, {}
for i, (st, en, s) in enumerate(rows):
    if s is None:
        continue
    dur.setdefault(s, []).append((en - st) / 1e3)
    if i + 1 < len(rows) and rows[i + 1][2] is not None:
        gap.setdefault(f"{s}->{rows[i + 1][2]}", []).append((rows[i + 1][0] - en) / 1e3)
pc = lambda x: dict(n=len(x), median=round(float(np.median(x)), 2), mean=round(float(np.mean(x)), 2),
                    p10=round(float(np.percentile(x, 10)), 2), p90=round(float(np.percentile(x, 90)), 2))
out = {"duration_us": {k: pc(v) for k, v in dur.items()}, "gap_us": {k: pc(v) for k, v in gap.items()}}
# one layer = logits start -> next logits start (median), when the logits kernels follow each other back to back
ls = [st for st, en, s in rows if s in ("logits2_kernel", "logits_kernel")]
d = np.diff(ls) / 1e3
d = d[d < 1000]
if len(d):
    out["layer_period_us"] = pc(d)
print(json.dumps(out, indent=1))
