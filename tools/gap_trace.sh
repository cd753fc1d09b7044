#!/bin/bash
# kernel-to-kernel gaps inside the token graph: rocprofv3 --kernel-trace of a short greedy decode, end(n) -> start(n + 1) by pair of kernel classes
# usage: tools/gap_trace.sh [outdir]   (on the GPU box)
cd "$(dirname "$0")/.."
R=$PWD; O=${1:-$R/gpurun_out/r06y}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/gapkt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/gapkt -- python $R/bench.py --steps 64 --warmup 8 --no-cpu-baseline --ref-steps 0 --prefill-chunks 0 --config2-steps 0 --profile-reps 0 > /dev/null 2>&1
cd $R
python - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
rows = []
for f in glob.glob(O + "/gapkt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
def cls(n):
    for k, v in (("k_att_t", "att"), ("k_attout_t", "attout"), ("k_ffn_rk_t", "ffn_rk"), ("k_ffnv_t", "ffn_v"), ("k_head", "head"), ("k_first", "first"), ("k_argmax", "argmax")):
        if k in n: return v
    return None
gaps = collections.defaultdict(list); durs = collections.defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    a, b = cls(n0), cls(n1)
    if a and b and s1 - e0 < 20000: gaps[(a, b)].append(s1 - e0)
for s, e, n in rows:
    if cls(n): durs[cls(n)].append(e - s)
with open(O + "/gap_trace.txt", "w") as fo:
    fo.write("# rocprofv3 --kernel-trace of bench.py --steps 64: gap = start(next) - end(prev) inside the token graph, ns; duration = end - start\n")
    for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
        v = sorted(v); fo.write("gap %-8s -> %-8s n %5d  median %6d  p10 %6d  p90 %6d\n" % (k[0], k[1], len(v), v[len(v) // 2], v[len(v) // 10], v[9 * len(v) // 10]))
    for k, v in durs.items():
        v = sorted(v); fo.write("dur %-8s n %5d  median %6d  p10 %6d  p90 %6d\n" % (k, len(v), v[len(v) // 2], v[len(v) // 10], v[9 * len(v) // 10]))
print(open(O + "/gap_trace.txt").read())
PY
