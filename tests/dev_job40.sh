set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03ap; mkdir -p $O
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU"
for M in glass metal glass+metal; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$M -- python3 tests/dev_mat_split.py $M > $O/stats_$M.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQ -d $O/pmc_$M --output-format csv -- python3 tests/dev_mat_split.py $M > $O/pmc_$M.log 2>&1
  tail -1 $O/stats_$M.log
done
