set -e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r03x; mkdir -p $O
S="GNXR_AB_SPP"; P="GNXR_AB_PASSES"; B="GNXR_TRACE_CHUNK"
V="c512::$S=32,$P=8 c128::$S=32,$P=8,$B=128 c256::$S=32,$P=8,$B=256 c1024::$S=32,$P=8,$B=1024 c2048::$S=32,$P=8,$B=2048 c512b::$S=32,$P=8"
python tests/dev_ab.py --workload cfg3 $V > $O/ab_cfg3.log 2>&1; cut -c1-170 $O/ab_cfg3.log
