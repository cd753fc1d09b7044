#!/bin/bash
# prefill bench + per-kernel durations of one chunk (rocprofv3 kernel trace)
cd "$(dirname "$0")/.."
R=$PWD
timeout 200 python tools/prefill_bench.py "$@" 2>&1 | tail -1
rm -rf gpurun_out/prof/seqt
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof/seqt -- python $R/tools/prefill_bench.py --chunks 2 "$@" > /dev/null 2>&1
cd $R
python - "$(find gpurun_out/prof/seqt -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'k_seq' in r['Kernel_Name'] or 'k_mm8_seq' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_seq_embed' in r['Kernel_Name']][-1]
one=rows[idx:]
L=sum('k_seq_wkv' in r['Kernel_Name'] for r in one)
layer=one[1+9*(L//2):1+9*(L//2)+9]
for r in layer: print(f"{r['Kernel_Name'][:38]:40s} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.2f} us")
print("head gemm", (int(one[-1]['End_Timestamp'])-int(one[-1]['Start_Timestamp']))/1e3, "us; chunk span", (int(one[-1]['End_Timestamp'])-int(one[0]['Start_Timestamp']))/1e3, "us")
PY
