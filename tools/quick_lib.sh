#!/bin/bash
# experiment library without a full rebuild: recompile ONE source with extra flags, link it with cached objects of the others
# usage: tools/quick_lib.sh <out.so> <source.hip> [flags...]      (tools/quick_lib.sh --base rebuilds the cache)
set -e
cd "$(dirname "$0")/.."
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=on -I include"
SRCS="engine lines xpass_a xpass_b zfused resident"
mkdir -p build_ab/base
if [ "$1" == "--base" ]; then
  for s in $SRCS; do /opt/rocm/bin/hipcc $F -c professad_amd/csrc/$s.hip -o build_ab/base/$s.o & done
  wait; exit 0
fi
OUT=$1; SRC=$2; shift 2
/opt/rocm/bin/hipcc $F "$@" -c professad_amd/csrc/$SRC.hip -o build_ab/exp_$$.o
OBJS=""; for s in $SRCS; do [ "$s" == "$SRC" ] && OBJS="$OBJS build_ab/exp_$$.o" || OBJS="$OBJS build_ab/base/$s.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $OUT
rm -f build_ab/exp_$$.o
