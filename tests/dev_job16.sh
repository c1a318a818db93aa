set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03p; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -60 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
S="GNXR_AB_SPP"; P="GNXR_AB_PASSES"
V="occ::$S=32,$P=8 fixed8::$S=32,$P=8,GNXR_GRID_OCC=0 occ2::$S=32,$P=8 fixed8b::$S=32,$P=8,GNXR_GRID_OCC=0"
python tests/dev_ab.py --workload cfg3 $V > $O/ab_cfg3.log 2>&1; cat $O/ab_cfg3.log
python tests/dev_ab.py --workload cfg4 $V > $O/ab_cfg4.log 2>&1; cat $O/ab_cfg4.log
for k in 1 2; do
python bench.py --workload cfg5 --no-cpu-baseline > $O/cfg5_occ.json 2>$O/cfg5.err; python -c "
import json; d=json.loads(open('$O/cfg5_occ.json').read().strip().splitlines()[-1]); print('occ', d['value'], d['ms_per_step'], d['roofline']['kernel_seconds'])"
GNXR_GRID_OCC=0 python bench.py --workload cfg5 --no-cpu-baseline > $O/cfg5_fixed.json 2>$O/cfg5.err; python -c "
import json; d=json.loads(open('$O/cfg5_fixed.json').read().strip().splitlines()[-1]); print('fixed8', d['value'], d['ms_per_step'], d['roofline']['kernel_seconds'])"
done
