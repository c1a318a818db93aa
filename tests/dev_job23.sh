set -e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r03w; mkdir -p $O
export GNXR_REGIONS=2
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS -d $O/a --output-format csv -- python3 tests/dev_ab.py --child 32 cfg3 2 > $O/a.log 2>&1 || tail -3 $O/a.log
python tests/dev_pmc_sum.py $O/a 2>/dev/null | grep "k_trace4\|k_shade" | cut -c1-600
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES -d $O/b --output-format csv -- python3 tests/dev_ab.py --child 32 cfg3 2 > $O/b.log 2>&1 || tail -3 $O/b.log
python tests/dev_pmc_sum.py $O/b 2>/dev/null | grep "k_trace4\|k_shade" | cut -c1-600
