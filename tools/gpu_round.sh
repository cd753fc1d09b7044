#!/bin/bash
# One GPU-box session that produces everything profiles/rNN/ holds: GPU tests, the full bench line (reference-kernel
# gate + CPU baseline legs included), rocprofv3 kernel-trace stats of the bench command, a separate PMC pass (FETCH_SIZE
# only, no trace domains), the prefill report with per-kernel statistics, the other model sizes, the N = 2 dry runs of the multi-GPU bench line.  usage: tools/gpu_round.sh [r06]
cd "$(dirname "$0")/.."
R=$PWD; TAG=${1:-r06}; O=$R/gpurun_out/$TAG
mkdir -p $O
export PYTHONUNBUFFERED=1
if [ -z "$SKIP_TESTS" ]; then
  timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -rs -rf 2>&1 | grep -v "^loading\|^n_layers\|^n_embed\|amdgpu.ids\|socket.cpp\|^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|pipeline transport up" | tail -80 > $O/pytest_gpu.log; tail -12 $O/pytest_gpu.log
fi
timeout 900 python bench.py 2>$O/bench7b_full.err | tail -1 > $O/bench7b_full.json; cut -c1-300 $O/bench7b_full.json
# the N > 1 line as the driver will run it, dry on the one GPU: 2 ranks, torch.distributed over gloo; (a) the engine's NATIVE schedule over the
# shared-memory RCCL stand-in (tests/fake_rccl.cpp), 14B by default; (b) the Python schedule over torch P2P ops
RWKV_BENCH_BACKEND=gloo RWKV_BENCH_ONE_DEVICE=1 RWKV_RCCL_LIB=$R/tests/_build/libfake_rccl.so timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 64 --warmup 8 --prefill-chunks 2 2>$O/bench_n2_dryrun_native.err | grep '^{"metric"' | tail -1 > $O/bench_n2_dryrun_native.json; echo "native dry run: exit $? $(wc -c < $O/bench_n2_dryrun_native.json) bytes"
RWKV_BENCH_BACKEND=gloo RWKV_BENCH_ONE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 16 --warmup 2 --model 1B5 --no-cpu-baseline 2>$O/bench_n2_dryrun_python.err | grep '^{"metric"' | tail -1 > $O/bench_n2_dryrun_python.json; echo "python dry run: exit $? $(wc -c < $O/bench_n2_dryrun_python.json) bytes"
bash tools/seq_trace.sh $O 2>&1 | tail -16
python - $O <<'PY'
import json, subprocess, sys, os
O = sys.argv[1]
out = {}
for m in ("169M", "1B5", "3B", "14B"):
    r = subprocess.run([sys.executable, "bench.py", "--model", m, "--steps", "256", "--warmup", "16", "--no-cpu-baseline", "--ref-steps", "0", "--config2-steps", "0"], capture_output=True, text=True, timeout=600)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        out[m] = dict(tokens_per_s=d["value"], ms_per_step=d["ms_per_step"], end_to_end=d["end_to_end"], kernels=d["kernels"], roofline=d["roofline"], prefill=d.get("prefill"))
        print(m, d["value"], d["end_to_end"], d.get("prefill", {}).get("tokens_per_s"))
    except Exception as e:
        out[m] = dict(error=str(e), stderr=r.stderr[-400:])
json.dump(out, open(os.path.join(O, "bench_sizes.json"), "w"), indent=1)
PY
cd /tmp && export TMPDIR=/tmp
rm -rf $O/kt $O/pmc
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 128 --warmup 8 --no-cpu-baseline --ref-steps 0 --prefill-chunks 0 --config2-steps 0 2>/dev/null | tail -1 > $O/bench7b_under_rocprof.json
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --ref-steps 0 --profile-reps 4 --prefill-chunks 0 --config2-steps 0 > /dev/null 2>&1
cd $R
python - "$O" <<'PY'
import csv, glob, json, sys, collections, shutil, os
O = sys.argv[1]
for f in glob.glob(O + "/kt/**/*kernel_stats.csv", recursive=True): shutil.copy(f, O + "/bench7b_kernel_stats.csv")
for f in glob.glob(O + "/kt/**/*domain_stats.csv", recursive=True): shutil.copy(f, O + "/bench7b_domain_stats.csv")
rows = list(csv.reader(open(O + "/bench7b_kernel_stats.csv")))
for r in rows[:10]: print([c[:60] for c in r[:6]])
alg = dict(att_kvr_wkv=3, att_out=1, ffn_rk=5, ffn_v=4)
names = {"k_att<": "att_kvr_wkv", "k_attout<": "att_out", "k_ffn_rk<": "ffn_rk", "k_ffnv<": "ffn_v", "k_head<": "head",
         "k_att_t<": "att_kvr_wkv", "k_attout_t<": "att_out", "k_ffn_rk_t<": "ffn_rk", "k_ffnv_t<": "ffn_v"}      # (row form | tile form, tile.hip.h)
agg = collections.defaultdict(list)
for f in glob.glob(O + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "FETCH_SIZE": continue
        for k, v in names.items():
            if k in r["Kernel_Name"]: agg[v].append(float(r["Counter_Value"]))
D, V = 4096, 50277
traffic = {}
with open(O + "/bench7b_pmc_fetch_size_summary.csv", "w") as fo:
    fo.write("kernel,dispatches,mean_FETCH_SIZE_KB,hbm_read_bytes_per_launch_corrected_x2,algorithmic_weight_bytes,ratio\n")
    for k, v in agg.items():
        mean = sum(v) / len(v); b = int(mean * 1024 * 2)
        a = V * D if k == "head" else alg[k] * D * D
        traffic[k] = b
        fo.write(f"{k},{len(v)},{mean:.1f},{b},{a},{b / a:.4f}\n")
print(open(O + "/bench7b_pmc_fetch_size_summary.csv").read())
import hashlib
sys.path.insert(0, os.getcwd())
import bench
json.dump({"7B": traffic, "decode_src_sha256": bench.decode_src_digest(), "_note": "HBM read bytes per launch = mean FETCH_SIZE [KB] x 1024 x 2 (gfx950 correction, MI355X_MICROARCH.md HBM section); "
           "own rocprofv3 --pmc FETCH_SIZE pass of `bench.py --steps 8 --warmup 2` (tools/gpu_round.sh)"}, open(O + "/hbm_traffic.json", "w"), indent=1)
PY
rm -rf $O/kt $O/pmc
# the same counter pass for the other BASELINE sizes (14B: the N > 1 line's model; 1B5: config 2), merged into hbm_traffic.json
bash tools/pmc_traffic.sh $TAG 14B | tail -8
bash tools/pmc_traffic.sh $TAG 1B5 | tail -8
# chunk path: HBM read and write traffic per weight pass (32-token chunk; 512-token prompt), one counter per pass -> prefill_traffic.json + per-kernel tables
cd $R
bash tools/prefill_traffic.sh $TAG 2>&1 | tail -30
cd $R
ls $O
