cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03af; mkdir -p $O
for cfg in "GNXR_REGIONS=8 GNXR_LOOP_LAG=1" "GNXR_REGIONS=3 GNXR_LOOP_LAG=6 GNXR_PIPE_CUT=1" "GNXR_REGIONS=2 GNXR_LOOP_LAG=4 GNXR_PIPE_CUT=7" "GNXR_REGIONS=1"; do
  n=$(echo $cfg | tr ' =' '__')
  env $cfg timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/tests_$n.log 2>&1
  echo "$cfg -> $(tail -1 $O/tests_$n.log)"
done
