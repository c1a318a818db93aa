set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03h; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "hlbvh or bvh" > $O/gpu_tests_bvh.log 2>&1 || { tail -60 $O/gpu_tests_bvh.log; exit 1; }
tail -3 $O/gpu_tests_bvh.log
GNXR_VERBOSE=1 python tests/dev_build_time.py > $O/build_time.log 2>&1 || tail -5 $O/build_time.log; grep -v "scene:" $O/build_time.log | tail -40
