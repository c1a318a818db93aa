set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03f; mkdir -p $O
S="GNXR_AB_SPP"; P="GNXR_AB_PASSES"; R="GNXR_REGIONS"; D="GNXR_TRACE_DUAL=1"
V="single::$S=64,$P=4,$R=2 dual::$S=64,$P=4,$R=2,$D dA:ab_libs/lib_dA.so:$S=64,$P=4,$R=2,$D dB:ab_libs/lib_dB.so:$S=64,$P=4,$R=2,$D dC:ab_libs/lib_dC.so:$S=64,$P=4,$R=2,$D dD:ab_libs/lib_dD.so:$S=64,$P=4,$R=2,$D single2::$S=64,$P=4,$R=2 r4k16::$S=16,$P=64,$R=4 r4k32::$S=32,$P=32,$R=4"
python tests/dev_ab.py --workload cfg3 $V > $O/ab_dual.log 2>&1; cat $O/ab_dual.log
