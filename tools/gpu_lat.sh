#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/lat tools/bin/variants/_tree
cp pfac_amd/lib/libpfac.so pfac_amd/lib/libpfac_gfx950.so tools/bin/variants/_tree/
export PFAC_LAT_KIB=4,64,1024,2048,4096,8192,16384,32768,65536,131072,262144,524288
for v in "$@"; do
  src=tools/bin/variants/$v; [ "$v" = tree ] && src=tools/bin/variants/_tree
  cp $src/libpfac.so $src/libpfac_gfx950.so pfac_amd/lib/
  echo "== $v"; timeout 600 python tools/small_input_latency.py 2>&1 | grep -E "naive|filter"
done | tee gpurun_out/lat/lat.txt
cp tools/bin/variants/_tree/* pfac_amd/lib/
