# Dev job: the randomised device-vs-oracle sweep under unusual settings of the path loop's scheduling knobs (every case must stay bit-identical)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/stress; mkdir -p $O
i=0
for K in "GNXR_LOOP_LAG=1" "GNXR_LOOP_LAG=5" "GNXR_PIPE_CUT=1" "GNXR_PIPE_CUT=6" "GNXR_SHADE_STREAMS=1" "GNXR_SHADE_BLOCKS_PER_CU=2" "GNXR_REGIONS=8" "GNXR_REGIONS=3 GNXR_LOOP_LAG=1 GNXR_PIPE_CUT=2" "GNXR_NO_PEER=1 GNXR_VOL_PACK=0"; do
  i=$((i+1))
  env $K timeout -k 10 600 python tests/dev_sweep.py $((20+i)) 60 > $O/sweep_$i.log 2>&1 || { echo "FAILED under $K"; tail -20 $O/sweep_$i.log; exit 1; }
  echo "$K: $(tail -1 $O/sweep_$i.log)"
done
