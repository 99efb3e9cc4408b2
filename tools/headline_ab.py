"""Same-box alternation of the headline step (and the grid's budget-2048 / window-32 calls) between libpkv builds: one bench.py
process per library and round (PKV_LIB), ms per step and the mean device us of every kernel.   python tools/headline_ab.py libA.so libB.so [rounds]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
rounds = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 3
for r in range(rounds):
    for lib in libs:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--only-gqa-extra", "--no-parity"],
                             env=dict(os.environ, PKV_LIB=os.path.abspath(lib)), capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(lib, "FAILED", out.stderr[-400:], flush=True)
            continue
        d = json.loads(line[0])
        print(os.path.basename(lib), d["ms_per_step"], {k: v["avg_us"] for k, v in d["roofline_kernels"].items() if k in ("logits", "finalize", "topk", "gather", "logits_gqa4", "finalize_gqa4", "topk_gqa4")},
              {"gqa_us_per_layer": (d.get("extras") or {}).get("unexpanded_gqa_us_per_layer")}, flush=True)
