#!/bin/bash
# experiment libraries against each other on ONE box, three alternations, whole-evaluation time only: tools/ab_libs3.sh "<bench args>" lib1.so lib2.so ...
ARGS=$1; shift
for r in 1 2 3; do
  for lib in "$@"; do
    OFDFT_LIB=$lib OFDFT_LIB_F32=$lib python bench.py --steps 30 --warmup 5 --no-cpu-baseline $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s' % sys.argv[1], d['ms_per_step'], d['value'], d['reference_check'] and d['reference_check']['ok'])" "$lib"
  done
done
