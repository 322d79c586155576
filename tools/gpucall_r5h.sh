TAG=${1:-r05h}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_f32.py tests/test_configs45_gpu.py -x -q -m gpu > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${TAG}_pytest.log
O=gpurun_out/${TAG}_pk.jsonl; : > $O
for rep in 1 2; do
  for v in default nopk; do
    if [ $v = default ]; then E="A=1"; else E="OFDFT_LIB_F32=build_ab/lib_${v}_f32.so"; fi
    echo "{\"variant\": \"$v\", \"rep\": $rep}" >> $O
    env $E timeout -k 10 200 python tools/shape_probe.py f32 cfg2 1024x128x1024 >> $O 2>/dev/null
    env $E timeout -k 10 200 python tools/shape_probe.py f32 256x256x256 255x255x255 >> $O 2>/dev/null
  done
done
python - "$O" <<'PY'
import json, sys
v = None
for line in open(sys.argv[1]):
    d = json.loads(line)
    if 'variant' in d:
        v = (d['variant'], d['rep']); continue
    print(v, d['shape'], d['terms'], d['ms'], {k: x for k, x in list(d['ps_per_point'].items())[:10]})
PY
