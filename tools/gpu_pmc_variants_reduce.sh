#!/bin/bash
# tools/gpu_pmc_variants_reduce.sh <workload> variant...  -- PMC instruction counts of the compacted-output kernel per variant (GPU box only)
cd "$(dirname "$0")/.."
w=$1; shift
mkdir -p tools/bin/variants/_tree; cp pfac_amd/lib/libpfac.so pfac_amd/lib/libpfac_gfx950.so tools/bin/variants/_tree/
for v in "$@"; do
  src=tools/bin/variants/$v; [ "$v" = tree ] && src=tools/bin/variants/_tree
  cp $src/libpfac.so $src/libpfac_gfx950.so pfac_amd/lib/
  echo "== $v"
  python3 tools/pmc_run.py --kernel pfac_scan_filter --tag reduce_${w}_$v --counters "GRBM_GUI_ACTIVE,SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_INSTS_VMEM_RD,SQ_LDS_IDX_ACTIVE,SQ_LDS_BANK_CONFLICT,SQ_WAVE_CYCLES" -- tools/reduce_driver.py $w 4 | grep -v "^#"
done
cp tools/bin/variants/_tree/* pfac_amd/lib/
