#!/bin/bash
# parity of the compacted-output path + its time on C3 / C5 / C2 (GPU box only)
cd "$(dirname "$0")/.."
timeout 1200 python -m pytest tests -x -q -m gpu -k "reduce or Reduce or compacted or host" 2>&1 | tail -4
for w in c3 c5 c2; do python3 tools/reduce_driver.py $w 6; done
python3 tools/pmc_run.py --kernel pfac_scan_filter --tag reduce_c3 --counters "GRBM_GUI_ACTIVE,SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_INSTS_VMEM_RD,SQ_LDS_IDX_ACTIVE,SQ_LDS_BANK_CONFLICT,SQ_WAVE_CYCLES" -- tools/reduce_driver.py c3 4 | grep -E "VALU|LDS|duration"
