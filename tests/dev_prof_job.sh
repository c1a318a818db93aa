set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r01l
python bench.py > gpurun_out/r01l/bench_full.json 2> gpurun_out/r01l/bench_full.err
echo "bench done"; tail -c 600 gpurun_out/r01l/bench_full.json
python bench.py --workload cfg5 > gpurun_out/r01l/bench_cfg5.json 2> gpurun_out/r01l/bench_cfg5.err
echo "cfg5 done"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01l/stats -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r01l/stats_bench.log 2>&1
echo "stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01l/stats5 -- python3 bench.py --workload cfg5 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r01l/stats5_bench.log 2>&1
echo "stats5 done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r01l/fetch --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/r01l/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r01l/write --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/r01l/write.log 2>&1
echo "pmc done"
python tests/dev_traffic.py gpurun_out/r01l/fetch gpurun_out/r01l/write gpurun_out/r01l/traffic.json
find gpurun_out/r01l -name "*.csv" | head -20
# two ranks sharing the one GPU of this box over gloo: rehearsal of the N>1 path of bench.py (the driver runs the real RCCL scaling bench)
GNXR_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r01l/bench_2rank_gloo.json 2> gpurun_out/r01l/bench_2rank_gloo.err
echo "2-rank rehearsal done"; tail -c 400 gpurun_out/r01l/bench_2rank_gloo.json
