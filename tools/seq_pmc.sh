#!/bin/bash
# PMC passes over the prefill GEMM (own runs, no trace domains): where do k_mm8_seq's cycles go?
cd "$(dirname "$0")/.."
R=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/prof/spmc$i
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/prof/spmc$i -- python $R/tools/prefill_bench.py --chunks 1 --layers 2 > /dev/null 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/prof/spmc*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_mm8_seq" in k or "k_ffn_rk" in k:
            agg[k[:24]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n=%d" % len(next(iter(d.values()))))
PY
