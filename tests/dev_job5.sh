set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03e; mkdir -p $O
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_ANY"
# PMC of the single-ray and the two-ray traversal kernel, 64 spp in 2 sub-passes of 32
export GNXR_REGIONS=2
rocprofv3 --kernel-trace --pmc $SQ -d $O/pmc_single --output-format csv -- python3 tests/dev_ab.py --child 32 cfg3 2 > $O/pmc_single.log 2>&1
python tests/dev_pmc_json.py $O/pmc_single $O/pmc_single.json --steps-profiled 1 | grep -E "k_trace" | cut -c1-400
GNXR_TRACE_DUAL=1 rocprofv3 --kernel-trace --pmc $SQ -d $O/pmc_dual --output-format csv -- python3 tests/dev_ab.py --child 32 cfg3 2 > $O/pmc_dual.log 2>&1
python tests/dev_pmc_json.py $O/pmc_dual $O/pmc_dual.json --steps-profiled 1 | grep -E "k_trace" | cut -c1-400
# kernel trace of the small-sub-pass configuration: per-launch durations
export GNXR_REGIONS=4
rocprofv3 --kernel-trace --output-format csv -d $O/kt_r4k16 -- python3 tests/dev_ab.py --child 16 cfg3 16 > $O/kt_r4k16.log 2>&1
python - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/r03e/kt_r4k16/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
t0=min(int(r['Start_Timestamp']) for r in rows)
out=open('gpurun_out/r03e/kt_r4k16_timeline.txt','w')
for r in rows[-700:]:
    n=r['Kernel_Name'].split('(')[0].replace('void gnxr::','')[:40]
    out.write(f"{(int(r['Start_Timestamp'])-t0)/1e3:12.1f} us  dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:9.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size',''))}  {n}\n")
PY
tail -5 gpurun_out/r03e/kt_r4k16_timeline.txt
