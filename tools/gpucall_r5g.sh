TAG=${1:-r05g}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_f32.py -x -q -m gpu -k "generic or odd or exact or chirp or extent or golden or profess or anchor or graph" > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${TAG}_pytest.log
timeout -k 10 300 python tools/shape_probe.py 255x255x255 129x135x127 53x53x53 > gpurun_out/${TAG}_shapes.jsonl 2> gpurun_out/${TAG}_shapes.err; echo "shapes rc=$?"
python - gpurun_out/${TAG}_shapes.jsonl <<'PY'
import json, sys
for line in open(sys.argv[1]):
    d = json.loads(line)
    print(d['shape'], d['ms'], d['ps_per_point'])
PY
timeout -k 10 300 python tools/latency_probe.py 53 > gpurun_out/${TAG}_lat.jsonl 2> gpurun_out/${TAG}_lat.err; echo "latency rc=$?"; cat gpurun_out/${TAG}_lat.jsonl | cut -c1-700
