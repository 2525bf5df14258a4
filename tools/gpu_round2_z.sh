#!/bin/bash
O=gpurun_out/r02z; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/mix.txt
cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep.so
for v in slim3 slim2; do
  cp tools/bin/variants/$v.so pfac_amd/lib/libpfac_gfx950.so
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_$v.txt 2>&1; echo "pytest $v rc $?" >> $O/pytest_$v.txt; tail -2 $O/pytest_$v.txt
done
for v in cur slim2 slim3 cur slim3; do
  cp tools/bin/variants/$v.so pfac_amd/lib/libpfac_gfx950.so
  echo "== $v" >> $O/mix.txt
  timeout 120 python tools/placement_mix.py 2>&1 | grep -v amdgpu.ids >> $O/mix.txt
done
V=tools/bin/variants
REPEAT=3 WL="c2 c5" timeout 900 tools/ab.sh $V/cur.so $V/slim2.so $V/slim3.so > $O/ab.txt 2>&1
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so
cat $O/mix.txt $O/ab.txt
