set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
V="default mw1:ab_libs/lib_mw1.so d4g3:ab_libs/lib_d4g3.so d3g3:ab_libs/lib_d3g3.so d4g2:ab_libs/lib_d4g2.so default2"
python tests/dev_ab.py --spp 64 --passes 4 --workload cfg3 $V > $O/ab_cfg3.log 2>&1; cat $O/ab_cfg3.log
python tests/dev_ab.py --spp 64 --passes 4 --workload cfg4 $V > $O/ab_cfg4.log 2>&1; cat $O/ab_cfg4.log
python bench.py --workload cfg5 --no-cpu-baseline > $O/cfg5_default.json 2>$O/cfg5_default.err; tail -c 400 $O/cfg5_default.json
GNXR_LIB=$PWD/ab_libs/lib_mw1.so python bench.py --workload cfg5 --no-cpu-baseline > $O/cfg5_mw1.json 2>$O/cfg5_mw1.err; tail -c 400 $O/cfg5_mw1.json
