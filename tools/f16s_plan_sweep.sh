#!/bin/bash
# On-device precision experiment (GPU box, repo root): the all-pixel parity tests of the f16s mode against the real reference's
# goldens (24-frame clip; 16-frame second clip with other weights) under a list of plan overrides (SAM2MI_F16S_PLAN,
# engine_core.hip f16s_plan_init) - prints the three north-star metrics per plan.
#     tools/f16s_plan_sweep.sh "<plan>" "<plan>" ...      ("-" = the built-in plan)
for plan in "$@"; do
  if [ "$plan" = "-" ]; then unset SAM2MI_F16S_PLAN; else export SAM2MI_F16S_PLAN="$plan"; fi
  line=$(python -m pytest tests/test_precise_gpu.py -q -s -k "(video_precise and f16s-8) or (second_clip and f16s)" 2>&1 | grep -E "video worst|second clip|passed|failed" | sed -e 's/\[parity\] //' -e 's/all pixels: //' | tr '\n' ' ')
  echo "[plan $plan] $line"
done
