#!/bin/bash
# chirp-z path A/B on ONE box: experiment builds (OFDFT_LIB) against each other, per-kernel-class times on odd grids
# usage: tools/bs_ab.sh "<shapes>" lib1.so lib2.so ...     ("default" = the in-tree library)
SHAPES=$1; shift
for r in 1 2; do
  for lib in "$@"; do
    if [ "$lib" == "default" ]; then python tools/shape_probe.py $SHAPES; else OFDFT_LIB=$lib python tools/shape_probe.py $SHAPES; fi 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); pp=d['ps_per_point']; print('%-36s %-16s %8.3f ms  chirp-z %7.1f ps/pt  (x %.1f y %.1f z %.1f)' % (sys.argv[1], 'x'.join(map(str,d['shape'])), d['ms'], sum(v for k,v in pp.items() if k.startswith('bluestein')), pp.get('bluestein_x',0), pp.get('bluestein_y',0), pp.get('bluestein_z',0)))" $lib
  done
done
