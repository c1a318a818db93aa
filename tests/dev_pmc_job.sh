# dev tool: extra SQ counter passes on the kernels of a workload (run through gpurun).  usage: bash tests/dev_pmc_job.sh <tag> [lib.so]
# (TA_* / TCP_* counters are left out on purpose: on this pool a `--pmc TA_TA_BUSY_sum ...` pass aborted in rocprofv3 and then hung
# until the silence watchdog killed the run.)
set -e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=${1:-x}; O=gpurun_out/pmc_$T; mkdir -p $O
if [ -n "$2" ]; then export GNXR_LIB=$GRAFT_REPO_ROOT/$2; fi
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_IFETCH SQ_INST_CYCLES_SALU -d $O/a --output-format csv -- $B > $O/a.log 2>&1 || tail -3 $O/a.log
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_INSTS_SALU -d $O/d --output-format csv -- $B > $O/d.log 2>&1 || tail -3 $O/d.log
for x in a d; do python tests/dev_pmc_sum.py $O/$x 2>/dev/null | grep "k_trace\|k_shade\|k_vol" | cut -c1-700; done
