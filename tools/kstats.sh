#!/bin/bash
# tools/kstats.sh <stats-build.so> [workload]  -- run one launch of a -DPFAC_STATS=1 build and show the per-block counters
cp "$1" pfac_amd/lib/libpfac_gfx950.so
python bench.py --worker pmc --workload ${2:-c3} --no-verify 2>&1 | grep STATS | sort | uniq | head -${LINES_MAX:-12}
