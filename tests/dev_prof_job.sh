# The measurement protocol of a round (dev tool, run through gpurun): GPU tests, the three bench workloads, rocprofv3 kernel stats,
# and the PMC passes (each in its own run, --kernel-trace only, as the pool requires).
# Usage: bash tests/dev_prof_job.sh <tag> <stages>   stages: any of  test bench stats pmc rehearsal  (default: all)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-r02a}
STAGES=${2:-"test bench stats pmc rehearsal"}
has() { case " $STAGES " in *" $1 "*) return 0;; *) return 1;; esac; }
O=gpurun_out/$T
mkdir -p $O
if has test; then
  python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
  tail -3 $O/gpu_tests.log
fi
if has bench; then
python bench.py > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "cfg3 done"; tail -c 1500 $O/bench_cfg3.json
python bench.py --workload cfg4 > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "cfg4 done"; tail -c 600 $O/bench_cfg4.json
python bench.py --workload cfg5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "cfg5 done"; tail -c 600 $O/bench_cfg5.json
fi
if has stats; then
for W in cfg3 cfg4 cfg5; do
  S=4; if [ $W = cfg5 ]; then S=1; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$W -- python3 bench.py --workload $W --steps $S --warmup 1 --no-cpu-baseline --no-also > $O/stats_$W.log 2>&1
  echo "stats $W done"
done
find $O -name "*kernel_stats.csv" | head
fi
if has pmc; then
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_ANY"
for W in cfg3 cfg4 cfg5; do
  rocprofv3 --kernel-trace --pmc $SQ -d $O/pmc_sq_$W --output-format csv -- python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/pmc_sq_$W.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch_$W --output-format csv -- python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/fetch_$W.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write_$W --output-format csv -- python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/write_$W.log 2>&1
  python tests/dev_traffic.py $O/fetch_$W $O/write_$W $O/traffic_$W.json
  X=""; if [ $W = cfg5 ]; then X="--width 512 --height 512 --spp-per-step 256"; fi
  python tests/dev_pmc_json.py $O/pmc_sq_$W $O/pmc_$W.json --workload $W $X --steps-profiled 3 --traffic $O/traffic_$W.json
  echo "pmc $W done"
done
fi
if has rehearsal; then
# two ranks sharing the one GPU of this box over gloo: rehearsal of bench.py's own rank launch (the driver runs the real RCCL scaling bench)
GNXR_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err
echo "2-rank rehearsal done"; tail -c 500 $O/bench_2rank_gloo.json
fi
