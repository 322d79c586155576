#!/bin/bash
# fp32 experiment libraries against the default fp32 build on ONE box: tools/ab_libs_f32.sh lib1.so lib2.so ...
run() { env $1 python bench.py --dtype f32 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-50s' % sys.argv[1], d['ms_per_step'], d['value'], {k: v['ms_per_eval'] for k, v in d['kernels'].items() if v['share'] > 0.03})" "$1"; }
for r in 1 2; do
  run "A=default"
  for lib in "$@"; do run "OFDFT_LIB_F32=$lib"; done
done
