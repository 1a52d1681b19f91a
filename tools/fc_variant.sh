#!/bin/bash
# An experimental build of ONE engine source beside the shipped library (the other objects are reused):
#   tools/fc_variant.sh <name> <source stem: engine_fc | engine_fc_split | engine_tile16 | engine_regtile> <-D flags...>
# -> climateparameterizations.jl_amd/libcolnde_<name>.so, for tools/ab_fc.py / tools/ab_bench.py (A/B in one process).  Never shipped, never timed by bench.py.
set -e
cd "$(dirname "$0")/../climateparameterizations.jl_amd/csrc"
NAME=$1; STEM=$2; shift; shift
EXTRA=""
[ "$STEM" = "engine_regtile" ] && EXTRA="-fno-slp-vectorize"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $EXTRA "$@" -c $STEM.hip -o _build/${STEM}_v_$NAME.o
OBJS=""
for s in api engine_tile16 engine_regtile engine_fc engine_fc_split column_ops comm; do
  if [ "$s" = "$STEM" ]; then OBJS="$OBJS _build/${STEM}_v_$NAME.o"; else OBJS="$OBJS _build/$s.o"; fi
done
/opt/rocm/bin/hipcc -shared --offload-arch=gfx950 -o ../libcolnde_$NAME.so $OBJS -ldl
echo built libcolnde_$NAME.so
