set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03r; mkdir -p $O
S="GNXR_AB_SPP"; P="GNXR_AB_PASSES"; B="GNXR_SHADE_BLOCKS_PER_CU"
V="b8::$S=32,$P=8 b64::$S=32,$P=8,$B=64 b128::$S=32,$P=8,$B=128 b256::$S=32,$P=8,$B=256 b1024::$S=32,$P=8,$B=1024 b32::$S=32,$P=8,$B=32"
python tests/dev_ab.py --workload cfg3 $V > $O/ab_cfg3.log 2>&1; cut -c1-170 $O/ab_cfg3.log
python tests/dev_ab.py --workload cfg4 $V > $O/ab_cfg4.log 2>&1; cut -c1-170 $O/ab_cfg4.log
