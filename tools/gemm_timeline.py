"""phase timeline of one GEMM of the chunk path (kind 0 K/V/R, 1 att_out, 2 ffn k/r, 3 ffn_v): python tools/gemm_timeline.py [kind] [model] [rows 32|64]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
kind = int(sys.argv[1]) if len(sys.argv) > 1 else 2
import numpy as np, torch                                                 # noqa: E402
from rwkv_cpp_accelerated_amd import engine, modelfile as mf              # noqa: E402

model = sys.argv[2] if len(sys.argv) > 2 else "7B"
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 32
os.environ["RWKV_TL_CLASS"] = str((20 if rows > 32 else 10) + kind)      # 10 + kind: a 32-row chunk, 20 + kind: a 64-row pass
L, D = mf.SHAPES[model]
L = min(L, 8)
m = engine.RWKV(resident=True)
m.loadTensors(L, D, mf.synthetic_tensors_torch(L, D, seed=0), maxGPT=rows)
m.forward([5] * rows, engine.MODE_GPT)
names = ["entry", "requests issued", "A image staged", "first batch done", "last weights multiplied", "end", "first batch multiplied (k_seq_gemm_b)"]
for rep in range(3):
    buf = m.debug_timeline(9).reshape(-1, 8, 8)[:512].astype(np.int64)
    live = buf[:, :, 0] > 0
    t0 = buf[:, :, 0][live].min()
    us = (buf - t0) / 100.0
    print(f"kind {kind} rows {rows} RWKV_SEQ_B={os.environ.get('RWKV_SEQ_B', 'default')} rep {rep}: workgroups {int(live.any(axis=1).sum())}, span {us[:, :, 5][buf[:, :, 5] > 0].max():.2f} us")
    for ph, nm in enumerate(names):
        v = us[:, :, ph][buf[:, :, ph] > 0]
        if v.size:
            print(f"  {nm:24s} min {v.min():7.2f}  mean {v.mean():7.2f}  max {v.max():7.2f}")
    e = us[:, :, 5].max(axis=1)[live.any(axis=1)]
    print("  workgroup end by K-slice (block % 8):", np.round([e[x::8].mean() for x in range(8)], 2))
m.close()
