#!/bin/bash
# tools/ab.sh <variant.so> ...  -- bench pre-built kernel modules against each other on the GPU box.
# Variants are built here (hipcc cross-compiles) into variants/<name>.so; the script swaps each one in
# as pfac_amd/lib/libpfac_gfx950.so and prints one compact line per workload.
WL=${WL:-"c3 c2"}
for so in "$@"; do
  cp "$so" pfac_amd/lib/libpfac_gfx950.so
  for w in $WL; do
    extra=""
    case $w in c5h) w=c5; extra="--perf-mode hash";; esac
    python bench.py --steps ${STEPS:-20} --warmup 3 --workload $w $extra --no-cpu-baseline $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$(basename $so .so)', d['config']['workload'][:24], d['value'], 'GB/s', r['kernel_ms_avg'], 'ms frac', round(r['frac'],3), 'exact', d['config']['bit_exact'], 'reduce', d.get('reduce_api',{}).get('value'))"
  done
done
