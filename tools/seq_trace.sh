#!/bin/bash
# prefill bench + per-kernel statistics of the chunked path (rocprofv3 kernel trace); usage: tools/seq_trace.sh <outdir> [prefill_bench args]
cd "$(dirname "$0")/.."
R=$PWD; O=$1; shift
mkdir -p $O
timeout 300 python tools/prefill_bench.py "$@" 2>/dev/null | tail -1 | tee $O/prefill7b.json
rm -rf $O/seqt
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/seqt -- python $R/tools/prefill_bench.py --chunks 4 "$@" > /dev/null 2>&1
cd $R
f=$(find $O/seqt -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then cp $f $O/prefill7b_kernel_stats.csv; python - $O/prefill7b_kernel_stats.csv <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'k_seq' in r['Name']]
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs'])):
    print(f"{r['Name'][:70]:72s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us  {100*float(r['TotalDurationNs'])/tot:5.1f} %")
PY
fi
rm -rf $O/seqt
