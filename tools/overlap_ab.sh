#!/bin/bash
# A/B of tuning knobs on one box (frames/s of the default bench, 3 steps each)
run() { echo -n "$1: "; env $1 python bench.py --steps 3 --no-cpu-baseline --no-roofline 2>&1 | tail -n 1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])"; }
for k in "$@"; do run "$k"; done
