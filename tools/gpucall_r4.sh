# GPU check of a change: the parity tiers that exercise the hot path (not the whole suite), then both bench lines (no CPU baseline).
# usage: bash tools/gpucall_r4.sh <tag> [full]
TAG=${1:-q}
mkdir -p gpurun_out
SEL="tests/test_gpu_parity.py tests/test_gpu_f32.py"
[ "$2" == "full" ] && SEL="tests"
timeout -k 10 900 python -m pytest $SEL -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_tests.log
tail -3 gpurun_out/${TAG}_tests.log
grep -q "rc=0" gpurun_out/${TAG}_tests.log || exit 9
for dt in f64 f32; do
  for r in 1 2; do
  timeout -k 10 300 python bench.py --dtype $dt --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_$dt.json 2> gpurun_out/${TAG}_bench_$dt.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_bench_$dt.json').read().strip().splitlines()[-1])
print('$dt', d['value'], d['ms_per_step'], d['reference_check'] and d['reference_check']['rel_dE'])
print({k:(v['ms_per_eval'], v.get('frac')) for k,v in d['kernels'].items() if v['share']>0.03})
PY
  done
done
