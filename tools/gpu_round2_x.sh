#!/bin/bash
O=gpurun_out/r02x; mkdir -p $O; export TMPDIR=/tmp; rm -rf $O/*
SETS=("TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum TCC_HIT_sum"
      "TCC_MISS_sum TCC_WRITEBACK_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum"
      "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
      "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU"
      "GRBM_GUI_ACTIVE TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum"
      "TCC_TAG_STALL_sum TCC_EA0_ATOMIC_sum TCC_NORMAL_WRITEBACK_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum")
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/s$i -o s -- python3 tools/placement_pmc.py > $O/s$i.log 2>&1
done
python3 tools/placement_pmc_report.py $O/s1 $O/s2 $O/s3 $O/s4 $O/s5 $O/s6 2>&1 | tee $O/report.txt
