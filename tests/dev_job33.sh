set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03ag; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -60 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
for k in 1 2; do
python bench.py --workload cfg5 --no-cpu-baseline > $O/cfg5_quads.json 2>$O/cfg5.err; python -c "
import json; d=json.loads(open('$O/cfg5_quads.json').read().strip().splitlines()[-1]); print('quads', d['value'], d['ms_per_step'], d['roofline']['kernel_seconds'])"
GNXR_NO_DENSITY_QUADS=1 python bench.py --workload cfg5 --no-cpu-baseline > $O/cfg5_plain.json 2>$O/cfg5.err; python -c "
import json; d=json.loads(open('$O/cfg5_plain.json').read().strip().splitlines()[-1]); print('plain', d['value'], d['ms_per_step'], d['roofline']['kernel_seconds'])"
done
