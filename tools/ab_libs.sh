#!/bin/bash
# several experiment libraries against the default on ONE box: tools/ab_libs.sh lib1.so lib2.so ...
run() { env $1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-44s' % sys.argv[1], d['ms_per_step'], d['value'], {k: v['ms_per_eval'] for k, v in d['kernels'].items() if v['share'] > 0.045})" "$1"; }
for r in 1 2; do
  run "A=default"
  for lib in "$@"; do run "OFDFT_LIB=$lib"; done
done
