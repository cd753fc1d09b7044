"""Where does the mean-to-slowest-workgroup spread of a decode launch come from?  
For each per-layer decode class (RWKV_TL_CLASS 1..4) the phase timeline of the middle layer's launch, several repetitions, reduced per XCD
(workgroup b runs on XCD b % 8): when the loader wave's last DMA had landed ("loader done", stamp 2 of wave 7), when the consumers' loops
ended, when the workgroup ended -- mean over the XCD's 32 workgroups and repetitions, and the slowest workgroup.  Run once per engine build
(RWKV_LIB=... for a variant, e.g. -DRWKV_TILE_XCDMAP=1: XCD x owns the contiguous channel blocks 32 x .. 32 x + 31 instead of every eighth).
usage: python tools/spread.py [model] [reps]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
model = sys.argv[1] if len(sys.argv) > 1 else "7B"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
if os.environ.get("RWKV_TL_CLASS") is None:
    for cls in (1, 2, 3, 4):
        subprocess.run([sys.executable, os.path.abspath(__file__), model, str(reps)], env=dict(os.environ, RWKV_TL_CLASS=str(cls)), check=False)
    sys.exit(0)
import numpy as np, torch                                                 # noqa: E402
from rwkv_cpp_accelerated_amd import engine, modelfile as mf              # noqa: E402

cls = int(os.environ["RWKV_TL_CLASS"])
L, D = mf.SHAPES[model]
L = min(L, 8)
m = engine.RWKV(resident=True); m.loadTensors(L, D, mf.synthetic_tensors_torch(L, D, seed=0))
for tk in (5, 6, 7):
    m.forward(tk)
ld, le, en = [], [], []
for rep in range(reps):
    buf = m.debug_timeline(9).reshape(-1, 8, 8)[:256].astype(np.int64)
    t0 = buf[:, :, 0][buf[:, :, 0] > 0].min()
    us = (buf - t0) / 100.0
    ld.append(us[:, 7, 2]); le.append(us[:, :7, 6].max(axis=1)); en.append(us[:, :, 7].max(axis=1))
ld, le, en = np.array(ld), np.array(le), np.array(en)                    # [rep][block]
name = {1: "k_att", 2: "k_attout", 3: "k_ffn_rk", 4: "k_ffnv"}[cls]
print(f"== {name} ({model}, decode_form {m.decode_form()}, lib {os.environ.get('RWKV_LIB', 'in-tree')}): span {en.max(axis=1).mean():.2f} us (mean over {reps} reps of the slowest workgroup's end), "
      f"mean end {en.mean():.2f}, mean loader done {ld.mean():.2f}, mean loop end {le.mean():.2f}")
for nm, a in (("loader done", ld), ("loop end", le), ("end", en)):
    per = [a[:, x::8] for x in range(8)]
    print(f"  {nm:12s} per XCD mean " + " ".join(f"{p.mean():6.2f}" for p in per) + "   | max " + " ".join(f"{p.max(axis=1).mean():6.2f}" for p in per)
          + f"   | XCD means spread {max(p.mean() for p in per) - min(p.mean() for p in per):.2f}, within-XCD std {np.mean([p.std(axis=1).mean() for p in per]):.2f}")
c = np.corrcoef(en)
print(f"  rep-to-rep correlation of per-workgroup end times: {np.round(c[0, 1:4], 2)};  span - mean end = {en.max(axis=1).mean() - en.mean():.2f} us")
m.close()
