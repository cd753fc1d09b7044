#!/bin/bash
# only the HBM-traffic pass of tools/gpu_round.sh: rocprofv3 --pmc FETCH_SIZE (own pass, counters only) over a short bench.py, per-kernel means x 2 (gfx950
# correction), written with the digest of the decode sources it profiled (merged into <out>/hbm_traffic.json under the model's key; a
# file with another digest is started over).  usage: tools/pmc_traffic.sh [r06] [7B|14B|...]
cd "$(dirname "$0")/.."
R=$PWD; TAG=${1:-r06}; MODEL=${2:-7B}; O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc -- python $R/bench.py --model $MODEL --steps 8 --warmup 2 --no-cpu-baseline --ref-steps 0 --profile-reps 4 --prefill-chunks 0 --config2-steps 0 > /dev/null 2>&1
cd $R
python - "$O" "$MODEL" <<'PY'
import csv, glob, json, sys, collections, os
O, MODEL = sys.argv[1], sys.argv[2]
alg = dict(att_kvr_wkv=3, att_out=1, ffn_rk=5, ffn_v=4)
names = {"k_att<": "att_kvr_wkv", "k_attout<": "att_out", "k_ffn_rk<": "ffn_rk", "k_ffnv<": "ffn_v", "k_head<": "head",
         "k_att_t<": "att_kvr_wkv", "k_attout_t<": "att_out", "k_ffn_rk_t<": "ffn_rk", "k_ffnv_t<": "ffn_v"}
agg = collections.defaultdict(list)
for f in glob.glob(O + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "FETCH_SIZE": continue
        for k, v in names.items():
            if k in r["Kernel_Name"]: agg[v].append(float(r["Counter_Value"]))
sys.path.insert(0, os.getcwd())
from rwkv_cpp_accelerated_amd import modelfile as mf
D, V = mf.SHAPES[MODEL][1], 50277
traffic = {}
SUMMARY = O + ("/bench7b_pmc_fetch_size_summary.csv" if MODEL == "7B" else f"/bench{MODEL}_pmc_fetch_size_summary.csv")
with open(SUMMARY, "w") as fo:
    fo.write("kernel,dispatches,mean_FETCH_SIZE_KB,hbm_read_bytes_per_launch_corrected_x2,algorithmic_weight_bytes,ratio\n")
    for k, v in agg.items():
        mean = sum(v) / len(v); b = int(mean * 1024 * 2)
        a = V * D if k == "head" else alg[k] * D * D
        traffic[k] = b
        fo.write(f"{k},{len(v)},{mean:.1f},{b},{a},{b / a:.4f}\n")
print(open(SUMMARY).read())
import bench
digest = bench.decode_src_digest()
out = {}
for cand in (O + "/hbm_traffic.json", os.path.join("profiles", os.path.basename(O), "hbm_traffic.json")):
    try:
        d = json.load(open(cand))
        if d.get("decode_src_sha256") == digest:
            out = d
            break
    except Exception:
        pass
out[MODEL] = traffic
out["decode_src_sha256"] = digest
out["_note"] = ("HBM read bytes per launch = mean FETCH_SIZE [KB] x 1024 x 2 (gfx950 correction, MI355X_MICROARCH.md HBM section); "
                "own rocprofv3 --pmc FETCH_SIZE pass of `bench.py --model M --steps 8 --warmup 2` per model (tools/pmc_traffic.sh / gpu_round.sh)")
json.dump(out, open(O + "/hbm_traffic.json", "w"), indent=1)
PY
rm -rf $O/pmc
