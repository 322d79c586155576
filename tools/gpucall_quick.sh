# quick GPU check of a change: full -m gpu tier, then the bench line (no CPU baseline).  usage: bash tools/gpucall_quick.sh <tag>
TAG=${1:-q}
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_tests.log
tail -4 gpurun_out/${TAG}_tests.log
grep -q "rc=0" gpurun_out/${TAG}_tests.log || exit 9
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['reference_check'] and d['reference_check']['rel_dE'])
print({k:(v['ms_per_eval'], v.get('frac')) for k,v in d['kernels'].items() if v['share']>0.015})
PY
