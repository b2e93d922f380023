#!/bin/bash
# Round profile recipe (run on the GPU box from the repo root):  [BENCH_ARGS="--precision f16"] tools/profile_round.sh <tag>   e.g. r03a
# Full bench line, rocprofv3 kernel stats with and without the encoder prefetch stream, and the two PMC passes (FETCH_SIZE /
# WRITE_SIZE in SEPARATE runs, counters only - never combined with trace domains) behind roofline.traffic.
# Outputs land under gpurun_out/<tag>_*; copy what is to be judged into profiles/ afterwards:
#   python tools/pmc_summary.py gpurun_out/<tag>_pmc_f/*/*counter_collection.csv gpurun_out/<tag>_pmc_w/*/*counter_collection.csv "<label>" profiles/pmc_traffic.json
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 5 ${BENCH_ARGS} > gpurun_out/${TAG}_bench_full.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -- python3 bench.py --steps 2 --warmup 1 ${BENCH_ARGS} --no-cpu-baseline --no-roofline --no-secondary > gpurun_out/${TAG}_prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof_noov -- python3 bench.py --steps 2 --warmup 1 ${BENCH_ARGS} --no-cpu-baseline --no-roofline --no-secondary --no-overlap > gpurun_out/${TAG}_prof_noov.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${TAG}_pmc_f -- python3 bench.py --steps 1 --warmup 0 --frames 16 ${BENCH_ARGS} --no-cpu-baseline --no-roofline --no-secondary --no-overlap > gpurun_out/${TAG}_pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${TAG}_pmc_w -- python3 bench.py --steps 1 --warmup 0 --frames 16 ${BENCH_ARGS} --no-cpu-baseline --no-roofline --no-secondary --no-overlap > gpurun_out/${TAG}_pmc_w.log 2>&1
# SQ utilisation counters (MFMA busy, wait / stall / issue split, LDS stalls and bank conflicts), two passes of 8 SQ slots + GRBM
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${TAG}_pmc_a -- python3 bench.py --steps 1 --warmup 0 --frames 16 ${BENCH_ARGS} --no-cpu-baseline --no-roofline --no-secondary --no-overlap > gpurun_out/${TAG}_pmc_a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${TAG}_pmc_b -- python3 bench.py --steps 1 --warmup 0 --frames 16 ${BENCH_ARGS} --no-cpu-baseline --no-roofline --no-secondary --no-overlap > gpurun_out/${TAG}_pmc_b.log 2>&1
python tools/pmc_util_summary.py $(ls gpurun_out/${TAG}_pmc_a/*/*counter_collection.csv | head -1) $(ls gpurun_out/${TAG}_pmc_b/*/*counter_collection.csv | head -1) 16 > gpurun_out/${TAG}_util.md 2>&1
python tools/pmc_summary.py $(ls gpurun_out/${TAG}_pmc_f/*/*counter_collection.csv | head -1) $(ls gpurun_out/${TAG}_pmc_w/*/*counter_collection.csv | head -1) "${TAG} (bench.py --frames 16 ${BENCH_ARGS} --no-overlap)" gpurun_out/${TAG}_pmc_traffic.json > gpurun_out/${TAG}_pmc_traffic.md 2>&1
# the counter CSVs are large (one row per dispatch and counter): keep the summaries
for d in ${TAG}_pmc_a ${TAG}_pmc_b ${TAG}_pmc_f ${TAG}_pmc_w; do find gpurun_out/$d -name "*counter_collection.csv" -delete; done
cat gpurun_out/${TAG}_util.md
tail -n 1 gpurun_out/${TAG}_bench_full.log | cut -c1-300
python tools/trace_gaps.py $(ls gpurun_out/${TAG}_prof_noov/*/*kernel_trace.csv | head -1) 300 > gpurun_out/${TAG}_gaps_noov.txt 2>&1
cat gpurun_out/${TAG}_gaps_noov.txt
# keep only the small summaries (the traces themselves are large)
for d in ${TAG}_prof ${TAG}_prof_noov; do find gpurun_out/$d -name "*kernel_trace.csv" -delete; done
ls gpurun_out/${TAG}_prof/*/ | head
