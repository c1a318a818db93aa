# dev tool: extra PMC passes on the traversal kernel (run through gpurun).  usage: bash tests/dev_pmc_job.sh <tag> [lib.so]
set -e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=${1:-x}; O=gpurun_out/pmc_$T; mkdir -p $O
if [ -n "$2" ]; then export GNXR_LIB=$GRAFT_REPO_ROOT/$2; fi
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_IFETCH SQ_INST_CYCLES_SALU -d $O/a --output-format csv -- $B > $O/a.log 2>&1 || tail -3 $O/a.log
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum -d $O/b --output-format csv -- $B > $O/b.log 2>&1 || tail -3 $O/b.log
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TA_FLAT_READ_WAVEFRONTS_sum -d $O/c --output-format csv -- $B > $O/c.log 2>&1 || tail -3 $O/c.log
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_INSTS_SALU -d $O/d --output-format csv -- $B > $O/d.log 2>&1 || tail -3 $O/d.log
for x in a b c d; do python tests/dev_pmc_sum.py $O/$x 2>/dev/null | grep "k_trace" | cut -c1-700; done
