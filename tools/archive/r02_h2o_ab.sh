#!/bin/bash
# H2O kernels at S = 32768: the library before this round's H2O rework (tools/_h2o_base.so, built by hand from commit
# d56ac47) against the current one.
mkdir -p gpurun_out
python tools/h2o_ab.py ${S:-32768} tools/_h2o_base.so pyramidkv_amd/libpkv.so 2>&1 | tee gpurun_out/h2o_ab.txt
