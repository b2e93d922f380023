cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_video_gpu.py tests/test_precise_gpu.py tests/test_outliers_gpu.py -m gpu -q -s > gpurun_out/s57_parity.log 2>&1; echo rc=$?
grep -c "\[parity\]" gpurun_out/s57_parity.log; tail -2 gpurun_out/s57_parity.log
