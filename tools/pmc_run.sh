#!/bin/bash
# usage: tools/pmc_run.sh <workload> <outdir>   -- rocprofv3 PMC passes over bench.py (GPU box only)
W=$1; OUT=$2; mkdir -p $OUT
export TMPDIR=/tmp
run() { # name, counters...
  name=$1; shift
  timeout 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o $name -- python3 bench.py --worker pmc --workload $W --no-verify > $OUT/$name.json 2> $OUT/$name.err
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INSTS_SMEM
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
