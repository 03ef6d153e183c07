#!/bin/bash
# Diagnostic: build a variant of libwavehip.so in which ONE source file is compiled with extra flags
#   bash tools/variant_lib.sh <name> <source file in csrc> [-D...]    -> examples/bin/libwavehip_<name>.so
# Run any tool against it with WAVEHIP_LIB=examples/bin/libwavehip_<name>.so (A/B of a compile-time switch).
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
C="$R/wave_fenics_amd/csrc"
NAME=$1; SRC=$2; shift 2
[ -f "$C/${SRC%.*}.o" ] || python -c "from wave_fenics_amd import build; build.build()"
mkdir -p "$R/examples/bin"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics "$@" -I "$R/include" -I "$C" \
  -c "$C/$SRC" -o "$R/examples/bin/variant_$NAME.o"
OBJS=$(ls "$C"/*.o | grep -v "/${SRC%.*}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/examples/bin/libwavehip_$NAME.so" $OBJS "$R/examples/bin/variant_$NAME.o" -ldl
echo "$R/examples/bin/libwavehip_$NAME.so"
