set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03an; mkdir -p $O
# frames larger than the headline: the library sizes sub-passes in paths (64 M), so the resident state does not grow with the frame
python bench.py --width 3840 --height 2160 --steps 2 --warmup 1 --no-also --no-cpu-baseline > $O/bench_4k.json 2> $O/bench_4k.err; echo "4K done"
python bench.py --width 7680 --height 4320 --spp-per-step 32 --steps 2 --warmup 1 --no-also --no-cpu-baseline > $O/bench_8k.json 2> $O/bench_8k.err; echo "8K done"
for f in 4k 8k; do tail -1 $O/bench_$f.json | python -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$f', round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],1), 'ms/step', {k:c.get(k) for k in ('passes_in_flight','sub_passes','path_state_GB','loop_iterations')})"; done
timeout -k 10 900 python tests/dev_sweep.py 7 320 > $O/sweep.log 2>&1 || { tail -20 $O/sweep.log; exit 1; }
tail -3 $O/sweep.log
timeout -k 10 600 python tests/dev_bigcheck.py > $O/bigcheck.log 2>&1 || { tail -20 $O/bigcheck.log; exit 1; }
cut -c1-200 $O/bigcheck.log
