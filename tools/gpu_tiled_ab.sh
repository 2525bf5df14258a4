#!/bin/bash
# tools/gpu_tiled_ab.sh <tag> variant...   -- tiled-kernel variants (tools/bin/variants/<name>, "tree" = pfac_amd/lib): hostile tests + naive bench lines
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p gpurun_out/$tag tools/bin/variants/_tree
cp pfac_amd/lib/libpfac.so pfac_amd/lib/libpfac_gfx950.so tools/bin/variants/_tree/
for v in "$@"; do
  src=tools/bin/variants/$v; [ "$v" = tree ] && src=tools/bin/variants/_tree
  cp $src/libpfac.so $src/libpfac_gfx950.so pfac_amd/lib/
  echo "== $v"
  timeout 900 python -m pytest tests/test_hostile.py -x -q -m gpu -s -k "snort_length or every_position" 2>&1 | grep -E "GB/s|passed|failed|Error|assert" | sed 's/dense-global[^,]*, //g; s/hash-global[^,]*, //g'
  for w in c3 c2 c5; do
    timeout 300 python bench.py --workload $w --variant naive --steps 5 --warmup 2 --pmc off --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['value'], d['ms_per_step'], d['config'].get('bit_exact'), d['config'].get('kernel_launched'))"
  done
done 2>&1 | tee gpurun_out/$tag/tiled_ab.txt
cp tools/bin/variants/_tree/* pfac_amd/lib/
timeout 300 python tools/small_input_latency.py 2>&1 | grep -E "naive|auto" | tee gpurun_out/$tag/latency.txt
