#!/bin/bash
# PFAC_TIMING build over the four buffer pairs of bench.py --spread: where does a scanning wave's time go on a fast and on a slow pair?
O=gpurun_out/r02tim2; mkdir -p $O; export TMPDIR=/tmp
cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep.so
cp tools/bin/variants/tim.so pfac_amd/lib/libpfac_gfx950.so
timeout 600 python bench.py --steps 3 --warmup 1 --workload c3 --no-cpu-baseline --no-other-configs --spread --pmc off 2> $O/err.txt > $O/out.json
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so
grep "scanners 14" $O/err.txt | sed 's/(\([0-9]*\) cyc\/wave)//g' | awk '{print NR": "$0}' | cut -c1-330 | tail -50
python3 -c "
import json; d=json.loads(open('$O/out.json').read().strip().splitlines()[-1]); print(d['roofline']['placement_spread_kernel_ms'])"
