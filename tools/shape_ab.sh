#!/bin/bash
# per-kernel-class A/B of experiment libraries on chosen shapes: tools/shape_ab.sh "<shapes>" default lib1.so ...
SHAPES=$1; shift
for r in 1 2; do
  for lib in "$@"; do
    if [ "$lib" == "default" ]; then python tools/shape_probe.py $SHAPES; else OFDFT_LIB=$lib python tools/shape_probe.py $SHAPES; fi 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); pp=d['ps_per_point']; print('%-34s %-14s %7.3f ms  ' % (sys.argv[1], 'x'.join(map(str,d['shape'])), d['ms']) + ' '.join('%s %.1f' % (k, v) for k, v in list(pp.items())[:9]))" $lib
  done
done
