#!/bin/bash
O=gpurun_out/r02v; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/mix.txt
cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep.so
for v in cur vg cur vg; do
  cp tools/bin/variants/$v.so pfac_amd/lib/libpfac_gfx950.so
  echo "== $v" >> $O/mix.txt
  timeout 120 python tools/placement_mix.py 2>&1 | grep -v amdgpu.ids >> $O/mix.txt
done
V=tools/bin/variants
REPEAT=3 WL="c2" timeout 900 tools/ab.sh $V/cur.so $V/vg.so > $O/ab.txt 2>&1
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so
cat $O/mix.txt $O/ab.txt
