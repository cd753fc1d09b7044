#!/bin/bash
# HBM traffic of the chunk path from counters: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (one counter per pass, counters only) over
# (a) BASELINE config 5, one 32-token chunk per call, and (b) a 512-token prompt in one call (64-row passes, three stage streams), summed over
# the k_seq_* kernels per weight pass and set against the weight bytes of a pass -> <out>/prefill_traffic.json (what bench.py's prefill /
# long_prompt legs quote, keyed on the digest of seq.hip.h + engine.hip) + the per-kernel tables.  usage: tools/prefill_traffic.sh [r06]
cd "$(dirname "$0")/.."
R=$PWD; TAG=${1:-r06}; O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for CFG in "chunk32 32 2" "prompt512 512 1"; do
  set -- $CFG; NAME=$1; TOK=$2; CH=$3
  for CTR in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmcs
    timeout 300 rocprofv3 --pmc $CTR --output-format csv -d $O/pmcs -- python $R/tools/prefill_bench.py --tokens $TOK --chunks $CH > /dev/null 2>&1
    python - "$O" $CTR $NAME $TOK $CH <<'PY'
import csv, glob, sys, collections, json, os
O, ctr, name, tok, ch = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
calls = ch + 1                                            # prefill_bench.py: one warm-up call + `chunks` timed calls
passes = calls * ((tok + 63) // 64)                       # weight passes made (a call of <= 32 rows: one 32-row pass)
agg = collections.defaultdict(list)
for f in glob.glob(O + "/pmcs/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == ctr and "k_seq" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0][:90]].append(float(r["Counter_Value"]))
with open(f"{O}/{name}_pmc_{ctr.lower()}.csv", "w") as fo:
    fo.write(f"kernel,dispatches,mean_{ctr}_KB,total_{ctr}_MB_per_weight_pass_uncorrected\n")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        fo.write(f"\"{k}\",{len(v)},{sum(v) / len(v):.1f},{sum(v) / 1024 / passes:.1f}\n")
tot_kb = sum(sum(v) for v in agg.values()) / passes
gemm_kb = sum(sum(v) for k, v in agg.items() if "gemm" in k) / passes
p = f"{O}/prefill_traffic_parts.json"
d = json.load(open(p)) if os.path.exists(p) else {}
d.setdefault(name, {})[ctr] = dict(kb_per_pass=tot_kb, gemm_kb_per_pass=gemm_kb, passes=passes, tokens=tok)
json.dump(d, open(p, "w"), indent=1)
print(name, ctr, f"{tot_kb / 1024:.1f} MB per weight pass (uncorrected), GEMMs {gemm_kb / 1024:.1f}")
PY
  done
done
rm -rf $O/pmcs
cd $R
python - "$O" <<'PY'
import json, sys, os
O = sys.argv[1]
sys.path.insert(0, os.getcwd())
import bench
from rwkv_cpp_accelerated_amd import modelfile as mf
L, D = mf.SHAPES["7B"]
parts = json.load(open(O + "/prefill_traffic_parts.json"))
out = {"seq_src_sha256": bench.seq_src_digest(), "7B": {},
       "_note": "per weight pass of the 7B chunk path, summed over the k_seq_* kernels (tools/prefill_traffic.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per "
                "run, counters only).  Reads = FETCH_SIZE x 2 (gfx950 correction of MI355X_MICROARCH.md, calibrated here on the GEMMs: their corrected reads = their "
                "weights + the activation images); writes = WRITE_SIZE x 1 (calibrated on the GEMMs' partial-value images: ffn k/r at 32 rows stores 8 x 1280 tiles x "
                "2 KiB = 21.0 MB and counts 21.0).  weight_bytes = 13 L D^2 + V D per 32-row half of the pass"}
for name, halves in (("chunk32", 1), ("prompt512", 2)):
    if name not in parts or "FETCH_SIZE" not in parts[name] or "WRITE_SIZE" not in parts[name]:
        continue
    w = 13 * L * D * D + halves * mf.VOCAB * D
    rd = parts[name]["FETCH_SIZE"]["kb_per_pass"] * 1024 * 2
    wr = parts[name]["WRITE_SIZE"]["kb_per_pass"] * 1024
    out["7B"][name] = dict(rows_per_pass=32 * halves, weight_bytes_per_pass=w, hbm_read_bytes_per_pass=int(rd), hbm_write_bytes_per_pass=int(wr),
                           read_over_weights=round(rd / w, 4), read_plus_write_over_weights=round((rd + wr) / w, 4),
                           gemm_read_bytes_per_pass=int(parts[name]["FETCH_SIZE"]["gemm_kb_per_pass"] * 2048), gemm_write_bytes_per_pass=int(parts[name]["WRITE_SIZE"]["gemm_kb_per_pass"] * 1024))
json.dump(out, open(O + "/prefill_traffic.json", "w"), indent=1)
print(json.dumps(out["7B"], indent=1))
PY
rm -f $O/prefill_traffic_parts.json
