#!/bin/bash
# Build libiunet.so for gfx950 (cross-compiles without a GPU).  build.sh: incremental (sources newer than their objects);
# build.sh --clean: every .hip file from scratch.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT" "$HERE/obj"
if [ "$1" = "--clean" ]; then rm -f "$HERE"/obj/*.o "$OUT/libiunet.so"; fi      # full rebuild (~30 s on 8 cores): what build() runs
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -Wno-int-to-pointer-cast"
pids=()
for f in "$HERE"/*.hip; do
  o="$HERE/obj/$(basename "${f%.hip}").o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$HERE/common.h" -nt "$o" ]; then
    hipcc $FLAGS -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libiunet.so" "$HERE"/obj/*.o
echo "built $OUT/libiunet.so"
