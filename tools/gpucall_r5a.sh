# round-5 iteration call: GPU test suite, then the config-5 kernels (fp32, 1024-point rows and lines) per class and the whole evaluation
TAG=${1:-r05a}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${TAG}_pytest.log
timeout -k 10 300 python tools/shape_probe.py f32 cfg2 1024x128x1024 1024x256x512 > gpurun_out/${TAG}_shapes_f32_cfg2.jsonl 2> gpurun_out/${TAG}_shapes_f32_cfg2.err; echo "shapes rc=$?"; cat gpurun_out/${TAG}_shapes_f32_cfg2.jsonl
timeout -k 10 400 python bench.py --dtype f32 --grid 1024 --cfg cfg2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_1024_f32_cfg2.json 2> gpurun_out/${TAG}_bench_1024_f32_cfg2.err; echo "bench 1024 rc=$?"
python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_bench_1024_f32_cfg2.json'))
print(d['ms_per_step'], {k:(v['ms_per_eval'], v.get('frac')) for k,v in d['kernels'].items()})
PY
