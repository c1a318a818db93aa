set -e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r03ac; mkdir -p $O
S="GNXR_AB_SPP"; P="GNXR_AB_PASSES"
V="half::$S=32,$P=8 l25:ab_libs/lib_l25.so:$S=32,$P=8 l35:ab_libs/lib_l35.so:$S=32,$P=8 l23:ab_libs/lib_l23.so:$S=32,$P=8 nospec:ab_libs/lib_nospec.so:$S=32,$P=8 half2::$S=32,$P=8"
python tests/dev_ab.py --workload cfg3 $V > $O/ab_cfg3.log 2>&1; cut -c1-170 $O/ab_cfg3.log
