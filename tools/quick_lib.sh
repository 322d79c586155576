#!/bin/bash
# experiment library without a full rebuild: recompile ONE source with extra flags, link it with cached objects of the others
# usage: [PREC=f32] tools/quick_lib.sh <out.so> <source.hip> [flags...]      ([PREC=f32] tools/quick_lib.sh --base rebuilds the cache)
set -e
cd "$(dirname "$0")/.."
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=on -I include"
BASE=build_ab/base
if [ "$PREC" == "f32" ]; then F="$F -DOFDFT_REAL_F32 -cl-single-precision-constant"; BASE=build_ab/base_f32; fi
SRCS="engine lines xpass_a xpass_b zfused resident"
mkdir -p $BASE
if [ "$1" == "--base" ]; then
  for s in $SRCS; do /opt/rocm/bin/hipcc $F -c professad_amd/csrc/$s.hip -o $BASE/$s.o & done
  wait; exit 0
fi
OUT=$1; SRC=$2; shift 2
/opt/rocm/bin/hipcc $F "$@" -c professad_amd/csrc/$SRC.hip -o build_ab/exp_$$.o
OBJS=""; for s in $SRCS; do [ "$s" == "$SRC" ] && OBJS="$OBJS build_ab/exp_$$.o" || OBJS="$OBJS $BASE/$s.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $OUT
rm -f build_ab/exp_$$.o
