#!/bin/bash
# tools/regs.sh [extra hipcc flags]  -- quick register / spill report of the bench instances (-DPFAC_QUICK build), ISA in /tmp/pfac_quick.s
cd "$(dirname "$0")/.."; mkdir -p /tmp/pfac_quick
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Ipfac_amd/csrc -DPFAC_QUICK "$@" -Rpass-analysis=kernel-resource-usage -save-temps=obj -c pfac_amd/csrc/scan_filter.hip -o /tmp/pfac_quick/q.o 2> /tmp/pfac_quick/ru.txt
cp /tmp/pfac_quick/scan_filter-hip-amdgcn-amd-amdhsa-gfx950.s /tmp/pfac_quick.s 2>/dev/null
grep -E "Function Name|SGPRs:|VGPRs:|Scratch|Spill" /tmp/pfac_quick/ru.txt | grep -A5 "pfac_scan_filter" | sed 's/.*remark: [^ ]* *//; s/ \[-Rpass.*//' | paste - - - - - - 
