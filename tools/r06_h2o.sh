set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/h2o; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "h2o" --timeout 600 > $O/pytest_h2o.txt 2>&1; echo "pytest exit $?" >> $O/pytest_h2o.txt
for S in 32768 8192; do for sc in 1 2 3 4 6 8; do echo "== S=$S H2O_SCALE=$sc"; H2O_SCALE=$sc timeout 200 python tools/h2o_ab.py $S pyramidkv_amd/libpkv.so; done; done > $O/h2o_ab.txt 2>&1
echo "== S=32768 H2O_OUTLIER=50" >> $O/h2o_ab.txt; H2O_OUTLIER=50 timeout 200 python tools/h2o_ab.py 32768 pyramidkv_amd/libpkv.so >> $O/h2o_ab.txt 2>&1
