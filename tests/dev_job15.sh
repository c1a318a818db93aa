set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03o; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -60 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
S="GNXR_AB_SPP"; P="GNXR_AB_PASSES"
V="esc::$S=32,$P=8 noesc::$S=32,$P=8,GNXR_NO_ESCAPE_QUEUE=1 esc2::$S=32,$P=8 noesc2::$S=32,$P=8,GNXR_NO_ESCAPE_QUEUE=1"
python tests/dev_ab.py --workload cfg4 $V > $O/ab_cfg4.log 2>&1; cat $O/ab_cfg4.log
