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
  stale=0
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ]; then stale=1; fi
  for h in "$HERE"/*.h; do                      # any shared header (common.h, pack_desc.h, x2_prep_desc.h ...) newer than the object
    if [ "$h" -nt "$o" ]; then stale=1; fi
  done
  if [ "$stale" = 1 ]; then
    hipcc $FLAGS -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libiunet.so" "$HERE"/obj/*.o
echo "built $OUT/libiunet.so"
