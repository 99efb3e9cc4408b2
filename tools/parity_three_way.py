#!/usr/bin/env python
"""Three-way same-chip parity (round-5 review item 1): libpkv (HIP) vs the reference's op sequence on the host CPU vs the same
op sequence on PyTorch-ROCm eager on this MI355X, on BASELINE configurations 2, 3 and 5 (tests/three_way.py has the method).

    python tools/parity_three_way.py            -> gpurun_out/parity_three_way.json (copied to profiles/rNN/ by the session)

config 2   PyramidKV budget 128, S = 8192, bf16 + fp16, all 32 layer budgets (234 ... 17)
config 3   SnapKV budgets 128 and 2048, S = 32768, bf16 + fp16 (H2O's eager form needs the 68.7 GB S x S tensor: CPU only, tests)
config 5   Ada-SnapKV (floor 0.2, normalize), Mistral GQA 32 / 8 heads, K/V un-expanded, S = 32768, budgets 128 and 2048
Also the `init_*` default knobs (window 32, avgpool-5, pyramidkv_utils.py:885-890) at S = 8192.
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import three_way as T3                 # noqa: E402
from inputs import make_qkv            # noqa: E402
from oracle import pkv_oracle as O     # noqa: E402
import pyramidkv_amd as P              # noqa: E402

W, D, H = 8, 128, 32


def main():
    t0 = time.time()
    out = {"device": torch.cuda.get_device_name(0), "torch": torch.__version__, "method": "tests/three_way.py",
           "host_threads": min(32, os.cpu_count() or 1), "configs": {}}
    for dt in ("bf16", "fp16"):
        # config 2: every pyramid layer budget on one set of tensors
        q, k, v = make_qkv(1, H, 8192, D, dt, "gauss", 6200)
        budgets = {}
        for layer in range(32):
            branch, kk = O.pyramid_budget(128, W, 32, layer, 8192)
            budgets["layer%02d" % layer] = kk
        out["configs"]["config2_pyramidkv_S8192_budget128_%s" % dt] = T3.window_policy(P, q, k, v, W, budgets)
        # the init_* defaults: window 32, avgpool-5 (pyramidkv_utils.py:885-890)
        out["configs"]["init_defaults_w32_avgpool5_S8192_%s" % dt] = T3.window_policy(
            P, q, k, v, 32, {"budget128": 128 - 32, "budget2048": 2048 - 32}, pooling="avgpool", kernel_size=5)
        # config 3: SnapKV at S = 32768, both budgets
        q, k, v = make_qkv(1, H, 32768, D, dt, "gauss", 6300)
        out["configs"]["config3_snapkv_S32768_%s" % dt] = T3.window_policy(P, q, k, v, W, {"budget128": 120, "budget2048": 2040})
        # config 5: Ada-SnapKV on Mistral's GQA layout
        q, k, v = make_qkv(1, H, 32768, D, dt, "gauss", 6500)
        ku, vu = k[:, ::4].contiguous(), v[:, ::4].contiguous()
        for cap in (128, 2048):
            out["configs"]["config5_adakv_gqa_S32768_budget%d_%s" % (cap, dt)] = T3.adakv(P, q, ku, vu, W, cap)
        print(dt, "done at %.0f s" % (time.time() - t0), flush=True)
    # the statements the suite asserts (tests/test_gpu_configs.py::test_three_way_*), evaluated over everything above
    summ = {"set_identity_at_budget128_every_pair": True, "budget2048_sequence_disagreement": {}}
    for name, rep in out["configs"].items():
        if "budgets" not in rep:
            continue
        for label, b in rep["budgets"].items():
            if b["k"] <= 234:
                for pair in ("hip_vs_cpu", "hip_vs_eager", "eager_vs_cpu"):
                    if b[pair]["set_rate"] != 1.0:
                        summ["set_identity_at_budget128_every_pair"] = False
                        summ.setdefault("set_differs", []).append([name, label, pair, b[pair]["identical_set"], b[pair]["heads"]])
            else:
                summ["budget2048_sequence_disagreement"][name + "/" + label] = {
                    p: b[p]["heads"] - b[p]["identical_sequence"] for p in ("hip_vs_cpu", "hip_vs_eager", "eager_vs_cpu")}
    out["summary"] = summ
    out["seconds"] = round(time.time() - t0, 1)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_three_way.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(summ))


if __name__ == "__main__":
    main()
