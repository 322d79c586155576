# A/B of the cross-wave x pass' cross-buffer count on ONE box (round 5): default build (one buffer for 1 -> 1 passes) against
# OFDFT_XC_ONEBUF=0 (two buffers everywhere, the round-4 kernel) and =2 (one buffer everywhere)
mkdir -p gpurun_out
O=gpurun_out/${1:-r05_onebuf}.jsonl; : > $O
for rep in 1 2; do
  for v in default ob0 ob2; do
    if [ $v = default ]; then E=""; else E="OFDFT_LIB_F32=build_ab/lib_${v}_f32.so OFDFT_LIB=build_ab/lib_${v}.so"; fi
    echo "{\"variant\": \"$v\", \"rep\": $rep}" >> $O
    env $E timeout -k 10 200 python tools/shape_probe.py f32 cfg2 1024x128x1024 >> $O 2>/dev/null
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
