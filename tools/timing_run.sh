#!/bin/bash
# tools/timing_run.sh <variant> [workloads...]  -- stage timing (PFAC_TIMING build, s_memtime at the stage boundaries) of one variant (GPU box only)
v=$1; shift; WL=${@:-c3 c2 c5}
cp pfac_amd/lib/libpfac.so /tmp/keep_libpfac.so; cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep_mod.so
cp tools/bin/variants/$v/*.so pfac_amd/lib/
for w in $WL; do echo "== $w"; python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --pmc off 2>&1 >/dev/null | grep -A1 "PFAC_TIMING blocks" | grep -v "scanners 16" | grep -A1 "PFAC_TIMING blocks" | tail -2; done
cp /tmp/keep_libpfac.so pfac_amd/lib/libpfac.so; cp /tmp/keep_mod.so pfac_amd/lib/libpfac_gfx950.so
