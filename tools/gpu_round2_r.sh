#!/bin/bash
O=gpurun_out/r02r; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/mix.txt
cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep.so
for v in cur rot1 rot5; do
  cp tools/bin/variants/$v.so pfac_amd/lib/libpfac_gfx950.so
  echo "== $v" >> $O/mix.txt
  timeout 120 python tools/placement_mix.py 2>&1 | grep -v amdgpu.ids >> $O/mix.txt
done
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so
cat $O/mix.txt
