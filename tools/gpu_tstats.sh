#!/bin/bash
# tiled-kernel step statistics (PFAC_TILED_STATS builds: tools/bin/variants/<name>, default tstats) on the hostile inputs and the bench streams (GPU box only)
cd "$(dirname "$0")/.."
mkdir -p tools/bin/variants/_tree; cp pfac_amd/lib/libpfac.so pfac_amd/lib/libpfac_gfx950.so tools/bin/variants/_tree/
for v in ${@:-tstats}; do
  echo "#### $v"
  cp tools/bin/variants/$v/*.so pfac_amd/lib/
  for c in snortlen allmatch; do echo "== $c"; python3 tools/hostile_driver.py $c naive 1 64 2>&1 | grep PFAC_TILED_STATS | tail -1; done
  for w in c3 c5 c2; do echo "== $w"; python3 bench.py --workload $w --variant naive --steps 1 --warmup 0 --pmc off --no-cpu-baseline --no-other-configs 2>&1 >/dev/null | grep PFAC_TILED_STATS | tail -1; done
done
cp tools/bin/variants/_tree/* pfac_amd/lib/
