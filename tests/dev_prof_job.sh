set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r01f
python bench.py > gpurun_out/r01f/bench_full.json 2> gpurun_out/r01f/bench_full.err
echo "bench done"; tail -c 600 gpurun_out/r01f/bench_full.json
python bench.py --workload cfg5 > gpurun_out/r01f/bench_cfg5.json 2> gpurun_out/r01f/bench_cfg5.err
echo "cfg5 done"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01f/stats -- python3 bench.py --steps 8 --warmup 1 --no-cpu-baseline > gpurun_out/r01f/stats_bench.log 2>&1
echo "stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01f/stats5 -- python3 bench.py --workload cfg5 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r01f/stats5_bench.log 2>&1
echo "stats5 done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r01f/fetch --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/r01f/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r01f/write --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/r01f/write.log 2>&1
echo "pmc done"
python tests/dev_traffic.py gpurun_out/r01f/fetch gpurun_out/r01f/write gpurun_out/r01f/traffic.json
find gpurun_out/r01f -name "*.csv" | head -20
