#!/bin/bash
# profile build: cycles per stage of the scanning waves (PFAC_TIMING), printed by the kernel module after every launch
O=gpurun_out/r02tim; mkdir -p $O; export TMPDIR=/tmp
cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep.so
cp tools/bin/variants/${1:-tim}.so pfac_amd/lib/libpfac_gfx950.so
for w in c3 c2 c5; do
  echo "== $w" >> $O/timing.txt
  timeout 600 python bench.py --steps 3 --warmup 1 --workload $w --no-cpu-baseline --no-other-configs --pmc off 2>&1 >/dev/null | grep PFAC_TIMING > $O/raw_$w.txt; grep "scanners 14" $O/raw_$w.txt | tail -2 >> $O/timing.txt; grep "scanners 16" $O/raw_$w.txt | tail -1 >> $O/timing.txt
done
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so
cat $O/timing.txt
