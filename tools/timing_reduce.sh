#!/bin/bash
# tools/timing_reduce.sh <variant> [workloads...]  -- stage timing of the compacted-output kernel (PFAC_TIMING build: the launches with 16 scanning waves) (GPU box only)
v=$1; shift; WL=${@:-c3}
cp pfac_amd/lib/libpfac.so /tmp/keep_libpfac.so; cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep_mod.so
cp tools/bin/variants/$v/*.so pfac_amd/lib/
for w in $WL; do echo "== $w"; python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --pmc off 2>&1 >/dev/null | grep "scanners 16" | tail -1 | tr ')' '\n' | sed 's/^ *//'; done
cp /tmp/keep_libpfac.so pfac_amd/lib/libpfac.so; cp /tmp/keep_mod.so pfac_amd/lib/libpfac_gfx950.so
