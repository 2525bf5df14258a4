#!/bin/bash
# tools/sweep.sh "<TPI> <SETS>" ...  -- rebuild the module with different tile/walker-set counts and bench (GPU box)
export PATH=/opt/rocm/bin:$PATH
for cfg in "$@"; do
  set -- $cfg
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Ipfac_amd/csrc -DPFAC_TILES_PER_ITER=$1 -DPFAC_WALK_SETS=$2 -shared -o pfac_amd/lib/libpfac_gfx950.so pfac_amd/csrc/scan_gfx950.hip 2>/dev/null
  for w in c3 c2; do
    python bench.py --steps 10 --warmup 2 --workload $w --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('TPI=$1 SETS=$2', d['config']['workload'][:2], d['value'], 'GB/s', d['roofline']['kernel_ms_avg'], 'ms exact', d['config']['bit_exact'])"
  done
done
