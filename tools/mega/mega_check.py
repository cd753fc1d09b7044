"""one-launch token (csrc/mega.hip.h) against the per-launch kernels: same tokens through two contexts of the same synthetic
model (RWKV_MEGA=0 / 1), logits and recurrent state compared; then a greedy chain.  usage: mega_check.py [L D] ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rwkv_cpp_accelerated_amd import engine, modelfile as mf

def ctx(L, D, t, mega):
    os.environ["RWKV_MEGA"] = "1" if mega else "0"
    m = engine.RWKV(resident=True)
    m.loadTensors(L, D, t, maxGPT=4)
    assert m.one_launch() == bool(mega), "one-launch mode not taken"
    return m

shapes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)] or [(2, 768), (3, 2048), (2, 4096), (2, 5120), (2, 2560)]
bad = 0
for L, D in shapes:
    t = mf.synthetic_tensors_torch(L, D, seed=3)
    a, b = ctx(L, D, t, False), ctx(L, D, t, True)
    worst = 0.0
    for step, tk in enumerate([5, 17, 50000, 1, 333, 9, 9, 4242]):
        la = np.array(a.forward(tk)[: mf.VOCAB], dtype=np.float64); lb = np.array(b.forward(tk)[: mf.VOCAB], dtype=np.float64)
        rel = np.abs(la - lb).max() / max(np.abs(la).max(), 1e-30)
        worst = max(worst, rel)
        if not np.isfinite(lb).all() or rel > 1e-5:
            print(f"  L={L} D={D} step {step}: rel {rel:.3e} finite={np.isfinite(lb).all()} argmax {la[1:].argmax()+1} vs {lb[1:].argmax()+1}")
            bad += 1
            break
    a.pull_state(1); b.pull_state(1)
    n = L * D
    srel = max(np.abs(x[:n] - y[:n]).max() / max(np.abs(x[:n]).max(), 1e-30) for x, y in zip(a.state.arrays(), b.state.arrays()))
    if not (srel < 1e-6): bad += 1
    ga = a.decode_greedy(7, 24); gb = b.decode_greedy(7, 24)
    same = list(ga) == list(gb)
    if not same: bad += 1
    print(f"L={L} D={D}: logits max rel {worst:.2e}  state rel {srel:.2e}  greedy 24 ids identical: {same}", flush=True)
    a.close(); b.close()
print("MEGA_CHECK", "FAIL" if bad else "OK")
sys.exit(1 if bad else 0)
