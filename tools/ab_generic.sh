# A/B of experiment libraries against the default on ONE box, per kernel class (tools/shape_probe.py):
#   bash tools/ab_generic.sh <tag> <variant name (build_ab/lib_<name>[_f32].so)> -- <shape_probe argument sets separated by ';'>
TAG=$1; V=$2; shift 3
mkdir -p gpurun_out
O=gpurun_out/${TAG}.jsonl; : > $O
IFS=';' read -ra SETS <<< "$*"
for rep in 1 2; do
  for v in default $V; do
    if [ $v = default ]; then E="A=1"; else E="OFDFT_LIB_F32=build_ab/lib_${v}_f32.so OFDFT_LIB=build_ab/lib_${v}.so"; fi
    echo "{\"variant\": \"$v\", \"rep\": $rep}" >> $O
    for s in "${SETS[@]}"; do env $E timeout -k 10 200 python tools/shape_probe.py $s >> $O 2>/dev/null; done
  done
done
python - "$O" <<'PY'
import json, sys
v = None
for line in open(sys.argv[1]):
    d = json.loads(line)
    if 'variant' in d:
        v = (d['variant'], d['rep']); continue
    print(v, d['shape'], d['dtype'][-7:], d['terms'], d['ms'], {k: x for k, x in list(d['ps_per_point'].items())[:9]})
PY
