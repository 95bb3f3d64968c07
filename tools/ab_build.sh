#!/bin/bash
# A/B library: rebuilds ONE source with extra -D flags and links it with the current objects into interactive-unet_amd/lib/libiunet_ab.so
# (select it with IUNET_LIB=<path>).   bash tools/ab_build.sh conv3_v4.hip -DV4_BW4
# The profiling switches that change results or paths (IUNET_V4_DBG, IUNET_F8K_DBG, IUNET_WG2_DBG) are compiled in only with -DIUNET_ABLATE:
#   bash tools/ab_build.sh conv3_f8k.hip -DIUNET_ABLATE   (the production library ignores those environment variables)
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
C=$R/interactive-unet_amd/csrc
src=$1; shift
mkdir -p /tmp/iunet_ab
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -Wno-int-to-pointer-cast "$@" -c $C/$src -o /tmp/iunet_ab/${src%.hip}.o
objs=""
for o in $C/obj/*.o; do b=$(basename $o); if [ "$b" = "${src%.hip}.o" ]; then objs="$objs /tmp/iunet_ab/$b"; else objs="$objs $o"; fi; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/interactive-unet_amd/lib/libiunet_ab.so $objs
echo built $R/interactive-unet_amd/lib/libiunet_ab.so
