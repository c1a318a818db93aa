set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03s; mkdir -p $O
S="GNXR_AB_SPP"; P="GNXR_AB_PASSES"; B="GNXR_GRID_BLOCKS_PER_CU"
V="g8::$S=32,$P=8 g16::$S=32,$P=8,$B=16 g32::$S=32,$P=8,$B=32 g64::$S=32,$P=8,$B=64 g8b::$S=32,$P=8"
python tests/dev_ab.py --workload cfg3 $V > $O/ab_cfg3.log 2>&1; cut -c1-170 $O/ab_cfg3.log
for b in 8 32; do
GNXR_GRID_BLOCKS_PER_CU=$b python bench.py --workload cfg5 --no-cpu-baseline > $O/cfg5_$b.json 2>$O/cfg5.err; python -c "
import json; d=json.loads(open('$O/cfg5_$b.json').read().strip().splitlines()[-1]); print('grid $b', d['value'], d['ms_per_step'], d['roofline']['kernel_seconds'])"
done
