#!/bin/bash
# Stall-attribution counters of the decode kernels (VERDICT r03 "make the floor claim a measurement").
# One rocprofv3 --pmc pass per counter group (SQ: 8 slots, TCC: 4; never combined with trace domains), each over
# `bench.py --steps 8 --profile-reps 4` of one model; means per dispatch and kernel class -> gpurun_out/<tag>/decode_pmc.csv
# usage: tools/pmc_decode.sh [tag=r04] [models="7B 1B5 14B"]
cd "$(dirname "$0")/.."
R=$PWD; TAG=${1:-r04}; MODELS=${2:-"7B 1B5 14B"}; O=$R/gpurun_out/$TAG
mkdir -p $O
export PYTHONUNBUFFERED=1
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P3="TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum"
P4="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"
P5="GRBM_GUI_ACTIVE GRBM_COUNT"
cd /tmp && export TMPDIR=/tmp
echo "model,pass,counters,status" > $O/decode_pmc_passes.csv
one_pass() {   # model, pass name, counters...
  local M=$1 P=$2; shift 2
  rm -rf $O/pmcd
  timeout 420 rocprofv3 --pmc "$@" --output-format csv -d $O/pmcd -- python $R/bench.py --model $M --steps 8 --warmup 2 --no-cpu-baseline --ref-steps 0 --profile-reps 4 --prefill-chunks 0 --config2-steps 0 > $O/pmcd.log 2>&1
  local n=$(find $O/pmcd -name '*counter_collection.csv' 2>/dev/null | wc -l)
  if [ "$n" -gt 0 ]; then
    python - "$O" "$M" "$P" <<'PY'
import csv, glob, sys, collections
O, M, P = sys.argv[1:4]
names = {"k_att<": "att_kvr_wkv", "k_attout<": "att_out", "k_ffn_rk<": "ffn_rk", "k_ffnv<": "ffn_v", "k_head<": "head", "k_first": "first",
         "k_attf<": "att_fused", "k_ffnf<": "ffn_fused"}
agg = collections.defaultdict(list)
for f in glob.glob(O + "/pmcd/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k, v in names.items():
            if k in r["Kernel_Name"]:
                agg[(v, r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(O + "/decode_pmc.csv", "a") as fo:
    for (k, c), v in sorted(agg.items()):
        fo.write(f"{M},{k},{c},{sum(v) / len(v):.1f},{len(v)}\n")
PY
    echo "$M,$P,\"$*\",ok" >> $O/decode_pmc_passes.csv
    return 0
  fi
  echo "$M,$P,\"$*\",FAILED: $(grep -i -m1 'error\|invalid\|not found\|unknown' $O/pmcd.log | cut -c1-160)" >> $O/decode_pmc_passes.csv
  return 1
}
[ -f $O/decode_pmc.csv ] || echo "model,kernel,counter,mean_per_dispatch,dispatches" > $O/decode_pmc.csv
for M in $MODELS; do
  for P in P1 P2 P3 P4 P5; do
    if [ "$M" != 7B ] && [ "$P" = P5 ]; then continue; fi
    C=${!P}
    if ! one_pass $M $P $C; then
      # a group with a counter this rocprofv3 does not know: collect its counters one by one
      for c in $C; do one_pass $M ${P}_$c $c; done
    fi
  done
done
rm -rf $O/pmcd
cat $O/decode_pmc_passes.csv
wc -l $O/decode_pmc.csv
