TAG=${1:-r05b}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${TAG}_pytest.log
O=gpurun_out/${TAG}_onebuf2.jsonl; : > $O
for rep in 1 2; do
  for v in default ob2; do
    if [ $v = default ]; then E="A=1"; else E="OFDFT_LIB_F32=build_ab/lib_${v}_f32.so OFDFT_LIB=build_ab/lib_${v}.so"; fi
    echo "{\"variant\": \"$v\", \"rep\": $rep}" >> $O
    env $E timeout -k 10 200 python tools/shape_probe.py f32 256x256x256 >> $O 2>/dev/null
    env $E timeout -k 10 200 python tools/shape_probe.py 256x256x256 512x256x256 >> $O 2>/dev/null
  done
done
python - "$O" <<'PY'
import json, sys
v = None
for line in open(sys.argv[1]):
    d = json.loads(line)
    if 'variant' in d:
        v = (d['variant'], d['rep']); continue
    print(v, d['shape'], d['dtype'][-7:], d['terms'], d['ms'], {k: x for k, x in d['ps_per_point'].items() if k.startswith('xfused')})
PY
timeout -k 10 400 python bench.py --dtype f32 --grid 1024 --cfg cfg2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_1024_f32_cfg2.json 2> gpurun_out/${TAG}_bench_1024_f32_cfg2.err; echo "bench 1024 rc=$?"
python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_bench_1024_f32_cfg2.json'))
print(d['ms_per_step'], {k:(v['ms_per_eval'], v.get('frac')) for k,v in d['kernels'].items()})
PY
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_256.json 2> gpurun_out/${TAG}_bench_256.err; echo "bench 256 rc=$?"
python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_bench_256.json'))
print(d['ms_per_step'], d['value'], d['reference_check'], {k:(v['ms_per_eval'], v.get('frac')) for k,v in d['kernels'].items()})
PY
