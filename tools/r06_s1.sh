set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s1; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python tools/parity_three_way.py > $O/three_way.log 2>&1; echo "exit $?" >> $O/three_way.log
cp gpurun_out/parity_three_way.json $O/ 2>/dev/null
timeout 300 python tools/merge_bench.py > $O/merge_bench.json 2> $O/merge_bench.err
for sc in 1 2 4 8; do echo "== H2O_SCALE=$sc"; H2O_SCALE=$sc timeout 200 python tools/h2o_ab.py 32768 pyramidkv_amd/libpkv.so; done > $O/h2o_ab.txt 2>&1
nproc > $O/nproc.txt; rocm-smi --showtopo > $O/topo.txt 2>&1
