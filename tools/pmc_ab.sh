#!/bin/bash
# tools/pmc_ab.sh <workload> <variant.so> ...   -- the same PMC sets for several kernel builds, side by side (GPU box only)
W=$1; shift
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
      "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD"
      "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL"
      "GRBM_GUI_ACTIVE GRBM_COUNT TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"
      "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum")
# (a set of TA_*/TD_* counters aborted rocprofv3 and hung the call for its whole limit: every run is under `timeout`)
export TMPDIR=/tmp
for so in "$@"; do
  name=$(basename $so .so); cp $so pfac_amd/lib/libpfac_gfx950.so
  OUT=gpurun_out/pmc_ab/$name; mkdir -p $OUT
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/s$i -o s$i -- python3 bench.py --worker pmc --workload $W --no-verify > $OUT/s$i.json 2> $OUT/s$i.err
  done
done
python3 - "$@" <<'PY'
import csv, glob, collections, sys, os
names = [os.path.basename(p)[:-3] for p in sys.argv[1:]]
table = collections.OrderedDict()
for n in names:
    for f in sorted(glob.glob("gpurun_out/pmc_ab/%s/*/*_counter_collection.csv" % n)):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "pfac_scan_filter" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            table.setdefault(k, {})[n] = sum(v) / len(v)
print("%-36s" % "counter" + "".join("%16s" % n[:15] for n in names))
for k, row in table.items():
    print("%-36s" % k + "".join("%16.4g" % row.get(n, float('nan')) for n in names))
PY
