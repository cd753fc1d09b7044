"""where the time of a one-launch token goes: per phase kind, the edge (last arrival -> poll -> hand-off loaded -> staged ->
first group) and the streaming span, from the kernel's own wall-clock stamps.  usage: mega_timeline.py [model] [layers]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rwkv_cpp_accelerated_amd import engine, modelfile as mf
model = sys.argv[1] if len(sys.argv) > 1 else "7B"
L, D = mf.SHAPES[model]
L = min(L, int(sys.argv[2]) if len(sys.argv) > 2 else 8)
t = mf.synthetic_tensors_torch(L, D, seed=0)
m = engine.RWKV(resident=True); m.loadTensors(L, D, t)
assert m.one_launch()
for tk in (5, 6, 7):
    m.forward(tk)
kinds = ["first", "att", "att_out", "ffn_rk", "ffn_v", "head"]
def kind_of(q, nq):
    return 0 if q == 0 else (5 if q == nq - 1 else 1 + (q - 1) % 4)
for rep in range(2):
    tl = m.mega_timeline(9).astype(np.int64)          # [wg][q][8]
    G, nq, _ = tl.shape
    t0 = tl[tl > 0].min()
    us = np.where(tl > 0, (tl - t0) / 100.0, np.nan)
    print(f"rep {rep}: token span {np.nanmax(us):.1f} us, {nq} phases")
    rows = {k: [] for k in range(6)}
    for q in range(1, nq):
        prev_arr = us[:, q - 1, 1]                    # arrival of each workgroup at the end of the previous phase
        last = np.nanmax(prev_arr)
        go, got, staged, first, arr = us[:, q, 2], us[:, q, 3], us[:, q, 4], us[:, q, 5], us[:, q, 1]
        l0, l1 = us[:, q, 6], us[:, q, 7]
        rows[kind_of(q, nq)].append([
            last - np.nanmean(prev_arr),              # arrival skew: mean workgroup waits this long for the last one
            np.nanmean(go) - last,                    # last arrival -> poll success
            np.nanmean(got - go),                     # -> tuple / partials loaded
            np.nanmean(staged - got),                 # -> staged
            np.nanmean(first - staged),               # -> first group landed (0: the ring was ahead)
            np.nanmean(arr - staged),                 # streaming span of the phase (staged -> workgroup arrives)
            np.nanmax(arr) - np.nanmax(prev_arr),     # phase period (last arrival to last arrival)
            np.nanmean(staged - l0),                  # how long before `staged` the loader had issued the phase's first unit
            np.nanmean(arr - l1),                     # last unit issued -> workgroup done
        ])
    print(f"  {'kind':8s} {'skew':>6s} {'->go':>6s} {'->data':>6s} {'->stgd':>6s} {'->1st':>6s} {'stream':>7s} {'PERIOD':>7s} {'lead':>6s} {'tail':>6s}")
    tot = 0.0
    for k in range(1, 6):
        if rows[k]:
            r = np.nanmean(np.array(rows[k]), axis=0)
            tot += r[6] * len(rows[k])
            print(f"  {kinds[k]:8s} " + " ".join(f"{v:6.2f}" for v in r[:5]) + f" {r[5]:7.2f} {r[6]:7.2f} {r[7]:6.2f} {r[8]:6.2f}")
    print(f"  first phase ends at {np.nanmax(us[:, 0, 1]):.2f} us; sum of periods {tot:.1f} us")
m.close()
