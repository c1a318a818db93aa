set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r01i
python bench.py > gpurun_out/r01i/bench_full.json 2> gpurun_out/r01i/bench_full.err
echo "bench done"; tail -c 600 gpurun_out/r01i/bench_full.json
python bench.py --workload cfg5 > gpurun_out/r01i/bench_cfg5.json 2> gpurun_out/r01i/bench_cfg5.err
echo "cfg5 done"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01i/stats -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r01i/stats_bench.log 2>&1
echo "stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01i/stats5 -- python3 bench.py --workload cfg5 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r01i/stats5_bench.log 2>&1
echo "stats5 done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r01i/fetch --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/r01i/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r01i/write --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/r01i/write.log 2>&1
echo "pmc done"
python tests/dev_traffic.py gpurun_out/r01i/fetch gpurun_out/r01i/write gpurun_out/r01i/traffic.json
find gpurun_out/r01i -name "*.csv" | head -20
