#!/bin/bash
# tools/ab.sh <variant.so> ...  -- bench pre-built kernel modules against each other on the GPU box.
# Variants are built in the container (tools/build_variants.sh) into tools/bin/variants/<name>.so; the script swaps each one in
# as pfac_amd/lib/libpfac_gfx950.so (and tools/bin/variants/libpfac_<name>.so as the host library, if present) and
# prints one line per workload: min / median kernel ms over REPEAT processes (run-to-run spread of one
# build is +-5 % on this pool, so single runs cannot rank variants).
WL=${WL:-"c3 c2"}; REPEAT=${REPEAT:-3}
for r in $(seq $REPEAT); do
  for so in "$@"; do
    name=$(basename $so .so)
    [ "$so" -ef pfac_amd/lib/libpfac_gfx950.so ] || cp "$so" pfac_amd/lib/libpfac_gfx950.so
    [ -f tools/bin/variants/libpfac_$name.so ] && cp tools/bin/variants/libpfac_$name.so pfac_amd/lib/libpfac.so
    for w in $WL; do
      extra=""; ww=$w
      case $w in c5h) ww=c5; extra="--perf-mode hash";; esac
      python bench.py --steps ${STEPS:-20} --warmup 3 --workload $ww $extra --no-cpu-baseline --no-other-configs --spread --pmc off $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
sp=list((r.get('placement_spread_kernel_ms') or {'x':r['kernel_ms_avg']}).values())
print('$name', '$w', r['kernel_ms_avg'], d['config']['bit_exact'], (d.get('reduce_api') or {}).get('ms_per_call'), min(sp), max(sp))" >> /tmp/ab_raw.txt
    done
  done
done
python - <<'PY'
import collections, statistics
rows = collections.OrderedDict()
for l in open('/tmp/ab_raw.txt'):
    n, w, ms, ok, red, lo, hi = l.split()
    rows.setdefault((n, w), []).append((float(ms), ok, None if red == 'None' else float(red), float(lo), float(hi)))
for (n, w), v in rows.items():
    ms = [x[0] for x in v]; red = [x[2] for x in v if x[2] is not None]
    print('%-14s %-4s kernel ms min %.4f median %.4f max %.4f  (n=%d) exact %s  reduce ms min %s  | 4 buffer pairs: fastest %.4f slowest %.4f' % (
        n, w, min(ms), statistics.median(ms), max(ms), len(ms), all(x[1] == 'True' for x in v), ('%.3f' % min(red)) if red else '-',
        min(x[3] for x in v), max(x[4] for x in v)))
PY
rm -f /tmp/ab_raw.txt
