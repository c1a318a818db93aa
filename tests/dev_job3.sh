set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03c; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
# 1024 spp per timed call: the steady state of the device-driven loop
S="GNXR_AB_SPP"; P="GNXR_AB_PASSES"; R="GNXR_REGIONS"
V="r2k128::$S=128,$P=8,$R=2 r4k32::$S=32,$P=32,$R=4 r4k16::$S=16,$P=64,$R=4 r6k16::$S=16,$P=64,$R=6 r8k8::$S=8,$P=128,$R=8 r6k8::$S=8,$P=128,$R=6 r4k16ng:ab_libs/lib_t4ng.so:$S=16,$P=64,$R=4 r2k128ng:ab_libs/lib_t4ng.so:$S=128,$P=8,$R=2 r4k16bcall:ab_libs/lib_bcall.so:$S=16,$P=64,$R=4 r4k16b::$S=16,$P=64,$R=4"
python tests/dev_ab.py --workload cfg3 $V > $O/ab_loop_cfg3.log 2>&1; cat $O/ab_loop_cfg3.log
V4="r4k16::$S=16,$P=16,$R=4 r4k16bcall:ab_libs/lib_bcall.so:$S=16,$P=16,$R=4 r4k16b::$S=16,$P=16,$R=4"
python tests/dev_ab.py --workload cfg4 $V4 > $O/ab_cfg4.log 2>&1; cat $O/ab_cfg4.log
