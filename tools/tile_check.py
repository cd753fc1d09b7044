"""tile-form decode kernels (csrc/tile.hip.h, env RWKV_TILE = bit mask of classes: 1 k_att, 2 k_attout, 4 k_ffn_rk, 8 k_ffnv) against the
row-form kernels on the same synthetic model (7B width by default): greedy ids and logits over a few teacher-forced tokens per mask, then (optional)
the phase timeline of one class in both forms.  python tools/tile_check.py [layers] [timeline class 1..4] [n_embed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch                                                 # noqa: E402
from rwkv_cpp_accelerated_amd import engine, modelfile as mf              # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
tlc = int(sys.argv[2]) if len(sys.argv) > 2 else 0
D = int(sys.argv[3]) if len(sys.argv) > 3 else 4096          # 4096: 16-row tiles (the chunk path's image); 5120 / 2048: 4-row tiles (decode-only image)
t = mf.synthetic_tensors_torch(L, D, seed=0)


def make(tile):
    os.environ["RWKV_TILE"] = str(tile)
    m = engine.RWKV(resident=True); m.loadTensors(L, D, t, maxGPT=32)
    return m


a = make(0)
ref, tk = [], 11
for step in range(24):
    la = a.forward(tk)[: mf.VOCAB].copy()
    ref.append((tk, la))
    tk = int(np.argmax(la[1:])) + 1
for mask in (1, 2, 4, 8, 15):
    b = make(mask)
    worst, same, fin = 0.0, True, True
    for tk, la in ref:
        lb = b.forward(tk)[: mf.VOCAB].copy()
        worst = max(worst, float(np.abs(la - lb).max() / np.abs(la).max()))
        same = same and int(np.argmax(la[1:])) == int(np.argmax(lb[1:]))
        fin = fin and bool(np.isfinite(lb).all())
    print(f"RWKV_TILE={mask:2d} vs row form, L={L}: 24 teacher-forced tokens, max rel logit diff {worst:.3e}, greedy ids identical: {same}, finite: {fin}", flush=True)
    if mask != 15:
        b.close()
if tlc:
    names = ["entry", "prologue issued", "loader done", "tuple arrived", "site reduced", "staged", "loop end", "end"]
    os.environ["RWKV_TL_CLASS"] = str(tlc)
    for label, m in ((f"class {tlc} row form", a), (f"class {tlc} tile form", b)):
        for rep in range(2):
            buf = m.debug_timeline(9).reshape(-1, 8, 8)[:256].astype(np.int64)
            t0 = buf[:, :, 0][buf[:, :, 0] > 0].min()
            us = (buf - t0) / 100.0
            print(f"{label} rep {rep}: kernel span {us[:, :, 7][buf[:, :, 7] > 0].max():.2f} us")
            for ph in range(8):
                v = us[:, :, ph][buf[:, :, ph] > 0]
                if v.size:
                    print(f"  {names[ph]:15s} min {v.min():6.2f}  mean {v.mean():6.2f}  max {v.max():6.2f}  (n={v.size})")
            print("  loop end by wave:", np.round([us[:, w, 6][buf[:, w, 6] > 0].mean() for w in range(8)], 2))
a.close(); b.close()
