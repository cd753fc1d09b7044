#!/bin/bash
# one GPU-box session: full GPU tests, golden fixtures from the reference kernel, bench, rocprof summaries
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/golden gpurun_out/prof
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^loading\|^n_layers\|^n_embed" | tail -4 > gpurun_out/pytest_gpu.log; cat gpurun_out/pytest_gpu.log
timeout 300 python tools/make_golden.py gpurun_out/golden 2>&1 | grep -v "^loading\|^n_layers\|^n_embed" | tail -6
timeout 600 python bench.py > gpurun_out/bench_full.log 2>gpurun_out/bench_full.err; tail -c 3000 gpurun_out/bench_full.log
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/kt -- python $R/bench.py --steps 128 --warmup 8 --no-cpu-baseline > $R/gpurun_out/prof/bench_under_rocprof.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof/pmc_fetch -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --profile-reps 4 > $R/gpurun_out/prof/bench_under_pmc.log 2>&1
cd $R
find gpurun_out/prof -name "*.csv" | head -20
python - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/prof/kt/**/*kernel_stats.csv", recursive=True):
    print(f); [print(r[:8]) for r in list(csv.reader(open(f)))[:12]]
for f in glob.glob("gpurun_out/prof/pmc_fetch/**/*counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(f, len(rows), list(rows[0].keys()) if rows else None)
    agg = collections.defaultdict(list)
    for r in rows:
        agg[r.get("Kernel_Name", "?")[:40]].append(float(r.get("Counter_Value", 0)))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print(f"  {k:40s} n={len(v)} mean={sum(v)/len(v):.1f}")
PY
