set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03q; mkdir -p $O
S="GNXR_AB_SPP"; P="GNXR_AB_PASSES"; B="GNXR_SHADE_BLOCKS_PER_CU"
V="b8::$S=32,$P=8 b4::$S=32,$P=8,$B=4 b12::$S=32,$P=8,$B=12 b16::$S=32,$P=8,$B=16 b32::$S=32,$P=8,$B=32 b64::$S=32,$P=8,$B=64 b8b::$S=32,$P=8"
python tests/dev_ab.py --workload cfg3 $V > $O/ab_cfg3.log 2>&1; cat $O/ab_cfg3.log
