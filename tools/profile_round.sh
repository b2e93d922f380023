#!/bin/bash
# Round profile recipe (run on the GPU box from the repo root): full bench line, rocprofv3 kernel stats with and without the
# encoder prefetch stream, and the two PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs) behind roofline.traffic.
# Outputs land under gpurun_out/; tools/pmc_summary.py + a copy of the *_kernel_stats.csv go to profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 5 > gpurun_out/r01d_bench_full.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01d_prof -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r01d_prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01d_prof_noov -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-overlap > gpurun_out/r01d_prof_noov.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r01d_pmc_f -- python3 bench.py --steps 1 --warmup 0 --frames 16 --no-cpu-baseline --no-roofline --no-overlap > gpurun_out/r01d_pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r01d_pmc_w -- python3 bench.py --steps 1 --warmup 0 --frames 16 --no-cpu-baseline --no-roofline --no-overlap > gpurun_out/r01d_pmc_w.log 2>&1
tail -n 1 gpurun_out/r01d_bench_full.log | cut -c1-300
ls gpurun_out/r01d_prof/*/ | head
