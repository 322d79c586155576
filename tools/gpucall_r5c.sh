TAG=${1:-r05c}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${TAG}_pytest.log
O=gpurun_out/${TAG}_zcx.jsonl; : > $O
for rep in 1 2; do
  for v in default zcx2; do
    if [ $v = default ]; then E="A=1"; else E="OFDFT_LIB_F32=build_ab/lib_${v}_f32.so"; fi
    echo "{\"variant\": \"$v\", \"rep\": $rep}" >> $O
    env $E timeout -k 10 200 python tools/shape_probe.py f32 cfg2 1024x128x1024 >> $O 2>/dev/null
    env $E timeout -k 10 200 python tools/shape_probe.py f32 256x256x256 >> $O 2>/dev/null
  done
done
python - "$O" <<'PY'
import json, sys
v = None
for line in open(sys.argv[1]):
    d = json.loads(line)
    if 'variant' in d:
        v = (d['variant'], d['rep']); continue
    print(v, d['shape'], d['dtype'][-7:], d['terms'], d['ms'], {k: x for k, x in d['ps_per_point'].items() if k.startswith('z')})
PY
