set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03aj; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -60 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
S="GNXR_AB_SPP"; P="GNXR_AB_PASSES"
V="new::$S=32,$P=8 old:ab_libs/lib_base.so:$S=32,$P=8 new2::$S=32,$P=8 old2:ab_libs/lib_base.so:$S=32,$P=8"
python tests/dev_ab.py --workload cfg3 $V > $O/ab_cfg3.log 2>&1; cut -c1-170 $O/ab_cfg3.log
python tests/dev_ab.py --workload cfg4 $V > $O/ab_cfg4.log 2>&1; cut -c1-170 $O/ab_cfg4.log
python bench.py --workload cfg5 --no-also --steps 3 --warmup 1 > $O/bench_cfg5.log 2>&1; cut -c1-400 $O/bench_cfg5.log
GNXR_LIB=$PWD/ab_libs/lib_base.so python bench.py --workload cfg5 --no-also --steps 3 --warmup 1 > $O/bench_cfg5_base.log 2>&1; cut -c1-400 $O/bench_cfg5_base.log
