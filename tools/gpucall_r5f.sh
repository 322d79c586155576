TAG=${1:-r05f}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "folded or fused_configs or big_scalars or bench_workload or graph_replay" > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${TAG}_pytest.log
O=gpurun_out/${TAG}_fold.jsonl; : > $O
for rep in 1 2; do
  for o in 1 0; do
    timeout -k 10 200 python tools/shape_probe.py opt28=$o 256x256x256 512x256x256 >> $O 2>/dev/null
    timeout -k 10 200 python tools/shape_probe.py f32 opt28=$o 256x256x256 >> $O 2>/dev/null
  done
done
python - "$O" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    d = json.loads(line)
    print(d['opts'], d['shape'], d['dtype'][-7:], d['ms'], {k: x for k, x in list(d['ps_per_point'].items())[:4]})
PY
