cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SAM2MI_MLP144_2WG=1 timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "mlp_fused" -s 2>&1 | grep -E "parity|passed|failed" | cut -c1-150
run() {
  timeout -k 10 300 env $2 python bench.py --steps 3 --no-cpu-baseline --no-secondary > gpurun_out/s58_$1.log 2>&1 || { echo "$1 failed"; tail -5 gpurun_out/s58_$1.log; return 1; }
  tail -n 1 gpurun_out/s58_$1.log | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
k=[ (n,v['avg_launch_us']) for n,v in d['roofline']['kernels'].items() if 'mlp_fused' in n]
print('$1', d['value'], d['config'].get('mask_checksum'), k)"
}
for rep in 1 2; do
run base_$rep X=1 || exit 1
run two_$rep SAM2MI_MLP144_2WG=1 || exit 1
done
