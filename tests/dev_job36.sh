set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03ak; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -60 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
S="GNXR_AB_SPP"; P="GNXR_AB_PASSES"
V="rec32::$S=32,$P=8 soa:ab_libs/lib_base.so:$S=32,$P=8 rec64:ab_libs/lib_rec64.so:$S=32,$P=8 rec32b::$S=32,$P=8 soab:ab_libs/lib_base.so:$S=32,$P=8 rec64b:ab_libs/lib_rec64.so:$S=32,$P=8"
python tests/dev_ab.py --workload cfg3 $V > $O/ab_cfg3.log 2>&1; cut -c1-170 $O/ab_cfg3.log
python tests/dev_ab.py --workload cfg4 $V > $O/ab_cfg4.log 2>&1; cut -c1-170 $O/ab_cfg4.log
for L in "" ab_libs/lib_base.so ab_libs/lib_rec64.so; do
  if [ -n "$L" ]; then export GNXR_LIB=$PWD/$L; fi
  python bench.py --workload cfg5 --no-also --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5', '$L', d['ms_per_step'], d['value'])"
done
