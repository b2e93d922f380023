cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r03g_gputests.log 2>&1; echo "gputests rc=$?"; tail -3 gpurun_out/r03g_gputests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03g_smoke.log 2>&1; echo "smoke rc=$?"
bash tools/profile_round.sh r03g > gpurun_out/r03g_round.log 2>&1; echo "round rc=$?"
bash tools/track_frame_profile.sh > gpurun_out/r03g_tf.log 2>&1; echo "tf rc=$?"
cp gpurun_out/track_frame.txt gpurun_out/r03g_track_frame.txt; cp gpurun_out/enc_batch.txt gpurun_out/r03g_encoder_batch.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03g_f16_prof_noov -- python3 bench.py --steps 2 --warmup 1 --precision f16 --no-cpu-baseline --no-roofline --no-secondary --no-overlap > gpurun_out/r03g_f16_prof_noov.log 2>&1; echo "f16 rc=$?"
find gpurun_out/r03g_f16_prof_noov -name "*kernel_trace.csv" -delete
timeout -k 10 200 python tools/event_timeline.py > gpurun_out/r03g_event_timeline.txt 2>&1; echo "ev rc=$?"
tail -n 1 gpurun_out/r03g_bench_full.log | cut -c1-160
tail -1 gpurun_out/r03g_track_frame.txt; tail -1 gpurun_out/r03g_encoder_batch.txt; grep "frames, wall" gpurun_out/r03g_event_timeline.txt
