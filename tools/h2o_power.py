"""Is the H2O pass limited by the chip's power budget?  For every library (PKV_LIB) and input kind, run the H2O score
at S = 32768, H = 32 in a loop for ~2.5 s while rocm-smi samples power and shader clock, and report per-pass times
(hipEvent, attached to the dispatches), mean power and mean sclk.
  python tools/h2o_power.py lib [lib ...]        (env: H2O_S = sequence length, H2O_KINDS = randn,zeros,ones, H2O_SECONDS = 2.5)"""
import json, os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import torch
    import pyramidkv_amd as P
    from pyramidkv_amd import _native as N
    S = int(os.environ.get("H2O_S", "32768"))
    g = torch.Generator(device="cuda").manual_seed(5)
    qr, kr = (torch.randn(1, 32, S, 128, device="cuda", generator=g).to(torch.bfloat16) for _ in range(2))
    samples = []
    stop = False

    def sampler():
        while not stop:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True)
            try:
                d = json.loads(r.stdout)
                c = next(iter(d.values()))
                pw = next((float(v) for k, v in c.items() if "Power" in k and "W" in k and re.match(r"^[0-9.]+$", str(v))), None)
                sc = next((v for k, v in c.items() if k.startswith("sclk")), None)
                m = re.search(r"(\d+)Mhz", str(sc))
                samples.append((pw, float(m.group(1)) if m else None))
            except Exception as e:       # noqa: BLE001
                samples.append((None, None))
            time.sleep(0.05)
    for kind in os.environ.get("H2O_KINDS", "randn,zeros,ones").split(","):
        if kind == "randn":
            q, k = qr, kr
        elif kind == "zeros":
            q, k = torch.zeros_like(qr), torch.zeros_like(kr)
        else:
            q, k = torch.ones_like(qr) * 0.25, torch.ones_like(kr) * 0.25
        for _ in range(2):
            P.ops.score_h2o(q, k, 8)
        torch.cuda.synchronize()
        samples.clear(); stop = False
        th = threading.Thread(target=sampler); th.start()
        N.prof_enable(True); N.prof_read(True)
        t0 = time.time(); n = 0
        while time.time() - t0 < float(os.environ.get("H2O_SECONDS", "2.5")):
            for _ in range(10):
                P.ops.score_h2o(q, k, 8)
            torch.cuda.synchronize(); n += 10
        wall = (time.time() - t0) / n
        stop = True; th.join()
        r = N.prof_read(True); N.prof_enable(False)
        ms = {kk: round(v[0] / max(1, v[1]), 3) for kk, v in r.items() if kk.startswith("h2o") and v[1]}
        pw = [s[0] for s in samples[2:] if s[0]]
        sc = [s[1] for s in samples[2:] if s[1]]
        print(json.dumps({"S": S, "data": kind, "ms": ms, "wall_ms_per_call": round(1e3 * wall, 2), "calls": n,
                          "power_w_mean": round(sum(pw) / len(pw), 1) if pw else None, "power_w_max": max(pw) if pw else None,
                          "sclk_mhz_mean": round(sum(sc) / len(sc)) if sc else None, "samples": len(samples)}), flush=True)
    sys.exit(0)
for lib in sys.argv[1:]:
    print("==", lib, flush=True)
    subprocess.run([sys.executable, __file__, "--one"], env=dict(os.environ, PKV_LIB=os.path.join(ROOT, lib)))
