set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03t; mkdir -p $O
S="GNXR_AB_SPP"; P="GNXR_AB_PASSES"; B="GNXR_SHADE_BLOCKS_PER_CU"
V="b8::$S=32,$P=8,$B=8 default32::$S=32,$P=8 b8b::$S=32,$P=8,$B=8 default32b::$S=32,$P=8 b64::$S=32,$P=8,$B=64"
python tests/dev_ab.py --workload cfg3 $V > $O/ab_cfg3.log 2>&1; cut -c1-170 $O/ab_cfg3.log
