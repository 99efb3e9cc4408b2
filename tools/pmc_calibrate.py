"""Calibrate rocprofv3's FETCH_SIZE on this box against kernels whose byte count is known (tools/bw_probe):
factor = true bytes / (FETCH_SIZE KiB * 1024) per load flavour.  Usage: pmc_calibrate.py <dir> <out.json>"""
import csv, glob, json, os, re, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "FETCH_SIZE":
            continue
        n = r["Kernel_Name"]
        m = re.match(r"(?:void )?(\w+)(<[^>]*>)?", n)
        key = (m.group(1) + (m.group(2) or ""), int(r["Grid_Size"]) if "Grid_Size" in r else int(r.get("Grid_Size_X", 0)))
        acc[key].append(float(r["Counter_Value"]) * 1024)
out = {}
for (name, grid), v in sorted(acc.items()):
    if name.startswith("read_kernel"):
        true = 1 << 30
    elif name.startswith("copy_kernel"):
        true = 1 << 30
    else:
        true = grid * 256          # one-pass kernels: one 256-B row per thread of the grid
    mean = sum(v) / len(v)
    out[f"{name} grid={grid}"] = {"true_read_bytes": true, "FETCH_SIZE_bytes": round(mean), "factor_true_over_counter": round(true / mean, 3) if mean else None,
                                  "launches": len(v)}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
