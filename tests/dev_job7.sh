set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03g; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -60 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
python tests/dev_build_time.py > $O/build_time.log 2>&1 || tail -5 $O/build_time.log; cat $O/build_time.log | tail -8
S="GNXR_AB_SPP"; P="GNXR_AB_PASSES"; R="GNXR_REGIONS"
V="r2k128::$S=128,$P=8,$R=2 r4k32::$S=32,$P=32,$R=4 r3k32::$S=32,$P=32,$R=3 r2k32::$S=32,$P=32,$R=2 r4k16::$S=16,$P=64,$R=4 r4k24::$S=24,$P=43,$R=4"
python tests/dev_ab.py --workload cfg3 $V > $O/ab_loop_cfg3.log 2>&1; cat $O/ab_loop_cfg3.log
