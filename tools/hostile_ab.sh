#!/bin/bash
# tools/hostile_ab.sh variant...  -- the two hostile-input tests (GB/s per kernel variant) under each library variant (GPU box only)
cd "$(dirname "$0")/.."
mkdir -p tools/bin/variants/_tree; cp pfac_amd/lib/libpfac.so pfac_amd/lib/libpfac_gfx950.so tools/bin/variants/_tree/
for v in "$@"; do
  src=tools/bin/variants/$v; [ "$v" = tree ] && src=tools/bin/variants/_tree
  cp $src/libpfac.so $src/libpfac_gfx950.so pfac_amd/lib/
  echo "== $v"
  PFAC_AB_OLD_LIBS=1 timeout 900 python -m pytest tests/test_hostile.py -x -q -m gpu -s -k "snort_length or every_position" 2>&1 | grep -E "GB/s|passed|failed|Error|assert" | sed 's/dense-global[^,]*, //g; s/hash-global[^,]*, //g' | cut -c1-600
done
cp tools/bin/variants/_tree/* pfac_amd/lib/
