set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03ai; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -60 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
S="GNXR_AB_SPP"; P="GNXR_AB_PASSES"
V="new::$S=32,$P=8 old:ab_libs/lib_lold.so:$S=32,$P=8 new2::$S=32,$P=8 old2:ab_libs/lib_lold.so:$S=32,$P=8"
python tests/dev_ab.py --workload cfg3 $V > $O/ab_cfg3.log 2>&1; cut -c1-170 $O/ab_cfg3.log
python tests/dev_ab.py --workload cfg4 $V > $O/ab_cfg4.log 2>&1; cut -c1-170 $O/ab_cfg4.log
