#!/bin/bash
# A/B of experiment libraries on the x-pass probe: tools/xc_variants.sh "<shapes and flags (f32)>" "<opts>" lib1.so lib2.so ...  (first: the default library)
SHAPES=$1; OPTS=$2; shift 2
fmt='
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d["shape"][0], d["xwave"], d["ms"], d["x_us"], "%.1e"%d["dgrad_rel"])'
for r in 1 2; do
  echo "== default"; timeout -k 10 200 python tools/xpass_ab.py opts=$OPTS $SHAPES 2>/dev/null | python -c "$fmt"
  for lib in "$@"; do
    echo "== $lib"; OFDFT_LIB=$lib OFDFT_LIB_F32=$lib timeout -k 10 200 python tools/xpass_ab.py opts=$OPTS $SHAPES 2>/dev/null | python -c "$fmt"
  done
done
