cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tf_prof -- python3 bench.py --steps 1 --warmup 1 --frames 40 --no-cpu-baseline --no-roofline --no-secondary --no-overlap > gpurun_out/tf_prof.log 2>&1
python tools/track_frame_trace.py $(ls gpurun_out/tf_prof/*/*kernel_trace.csv | head -1) 3 > gpurun_out/track_frame.txt 2>&1
python tools/track_frame_trace.py $(ls gpurun_out/tf_prof/*/*kernel_trace.csv | head -1) 2 enc > gpurun_out/enc_batch.txt 2>&1
find gpurun_out/tf_prof -name "*kernel_trace.csv" -delete
tail -3 gpurun_out/track_frame.txt
