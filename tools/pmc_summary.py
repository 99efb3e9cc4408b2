"""Summarise rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE) per pkv kernel -> profiles/pmc_traffic.json.
Units/corrections per MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of a wide coalesced streaming read, so the read side is doubled."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
dst = sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if "pkv::" not in name:
            continue
        short = name.split("pkv::")[1].split("_kernel")[0]
        pass_dir = os.path.basename(os.path.dirname(os.path.dirname(f)))
        two_tiles = short == "logits2" and (", 2, " in name or ",2," in name)      # logits2_kernel<T, 2, NT>: 17..32 columns
        if short == "logits2":      # the pipelined variant of the logits kernel reports under the same profiler id
            short = "logits"
        if pass_dir.startswith("pmc_gqa"):
            # bench.py --only-gqa-extra: the headline kernels run as well; only the un-expanded-K scan (two column tiles) is new
            if not two_tiles:
                continue
            short = "logits_gqa4"
        elif two_tiles:
            continue
        if pass_dir.startswith("pmc_gather"):
            # tools/gather_pmc.py: budget-2048 gather, B = 1 (512 workgroups of 256 threads) and B = 8 (4096)
            if short != "gather":
                continue
            wgs = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
            short = {512: "gather_cap2048_B1", 4096: "gather_cap2048_B8"}.get(wgs)
            if short is None:
                continue
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py --steps 1 --warmup 1",
       "correction": "FETCH_SIZE x2 (gfx950 wide-read undercount), KiB -> bytes", "kernels": {}}
for kname, ctr in acc.items():
    fetch = ctr.get("FETCH_SIZE", [])
    write = ctr.get("WRITE_SIZE", [])
    if not fetch and not write:         # directories of other counter passes (SQ_* of the H2O kernels)
        continue
    fe = sum(fetch) / len(fetch) * 1024 * 2 if fetch else None
    wr = sum(write) / len(write) * 1024 if write else None
    out["kernels"][kname] = {"fetch_bytes_per_launch_corrected": fe, "write_bytes_per_launch": wr,
                             "hbm_bytes_per_launch": (fe or 0) + (wr or 0) if (fe is not None or wr is not None) else None,
                             "launches_fetch": len(fetch), "launches_write": len(write)}
# identity of the kernel sources (the same comment-insensitive hash bench.py computes): the bench line only attaches traffic
# that belongs to them
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_src_sha16  # noqa: E402
out["kernel_src_sha16"] = kernel_src_sha16()
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
