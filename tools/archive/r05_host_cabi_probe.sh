#!/bin/bash
# How often does the C++ example die in teardown (heap check at exit), and does the RCCL section matter?
# One failure in six sessions so far: "corrupted size vs. prev_size in fastbins" after "host_cabi: ok".
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/hostcabi; mkdir -p $out; : > $out/runs.txt
run() { # label, n, cmd...
  local label=$1 n=$2; shift 2; local fails=0
  for i in $(seq 1 $n); do
    timeout 120 "$@" > $out/last.out 2> $out/last.err; rc=$?
    if [ $rc -ne 0 ]; then fails=$((fails+1)); echo "--- $label run $i rc=$rc" >> $out/fail_detail.txt; tail -3 $out/last.out >> $out/fail_detail.txt; tail -5 $out/last.err >> $out/fail_detail.txt; fi
  done
  echo "$label: $fails failures of $n" | tee -a $out/runs.txt
}
run prev 14 examples/host_cabi_prev
run new 14 examples/host_cabi
run new_no_rccl 14 examples/host_cabi --no-rccl
MALLOC_CHECK_=3 run prev_malloc_check 6 examples/host_cabi_prev
MALLOC_CHECK_=3 run new_malloc_check 6 examples/host_cabi
cat $out/runs.txt
