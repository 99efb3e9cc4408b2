"""Register / LDS / occupancy table of the kernels of one libpkv source file, as hipcc reports them
(-Rpass-analysis=kernel-resource-usage; no GPU needed).

  python tools/kernel_resources.py pkv_h2o.hip [-DH2O_LB=3 ...] [--filter BF16ELi4] [--asm out.s]
"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pyramidkv_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]


def resources(src, extra=(), asm=None):
    if os.path.basename(src) == "pkv_h2o.hip":
        extra = ["-fno-slp-vectorize", *extra]
    out = asm or "/tmp/_kres.s"
    r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                        os.path.join(CSRC, src), "-o", out], capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-3000:])
    rows, cur = [], None
    for ln in r.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", ln)
        if not m:
            continue
        t = m.group(1)
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.rsplit(":", 1)
            cur[k.strip()] = v.strip()
    return rows


def demangle(n):
    r = subprocess.run(["c++filt", n], capture_output=True, text=True)
    return r.stdout.strip() or n


if __name__ == "__main__":
    args = sys.argv[1:]
    flt = None
    asm = None
    if "--filter" in args:
        i = args.index("--filter"); flt = args[i + 1]; del args[i:i + 2]
    if "--asm" in args:
        i = args.index("--asm"); asm = args[i + 1]; del args[i:i + 2]
    src, extra = args[0], args[1:]
    print(f"# {src} {' '.join(extra)}")
    print("| kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | waves/SIMD | LDS static |")
    print("|---|---|---|---|---|---|---|")
    for r in resources(src, extra, asm):
        if flt and flt not in r["name"]:
            continue
        print(f"| `{demangle(r['name'])}` | {r.get('VGPRs')} | {r.get('AGPRs')} | {r.get('TotalSGPRs')} | {r.get('ScratchSize [bytes/lane]')} | "
              f"{r.get('Occupancy [waves/SIMD]')} | {r.get('LDS Size [bytes/block]')} |")
