import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from rwkv_cpp_accelerated_amd import engine, modelfile as mf
import oracle_lib

def run(L, D, gen, mode, maxGPT, toks):
    if gen == "torch":
        t = mf.synthetic_tensors_torch(L, D, seed=3)
        for i, x in enumerate(t):
            if x is not None and x.dtype != torch.uint8:
                if not torch.isfinite(x).all():
                    print("  non-finite input tensor", i, mf.NAMES[i])
    else:
        t = mf.synthetic_tensors(L, D, seed=3)
    m = engine.RWKV(resident=True); m.loadTensors(L, D, t, maxGPT=maxGPT)
    out = m.forward(toks, mode)[: len(toks) * mf.VOCAB].copy()
    print(f"L={L} D={D} gen={gen} mode={mode} maxGPT={maxGPT}: finite={np.isfinite(out).all()} absmax={np.nanmax(np.abs(out)):.3f}")
    if gen == "torch" and D <= 1024:
        host = [None if x is None else x.cpu().numpy() for x in t]
        om = oracle_lib.Oracle().from_tensors(L, D, host)
        ref = om.forward(toks, om.new_state(slots=maxGPT), mode=mode)
        print("   vs oracle max abs diff", np.abs(out.reshape(ref.shape) - ref).max(), "ref absmax", np.abs(ref).max())
    m.close()

run(2, 768, "torch", 1, 1, [7])
run(2, 768, "torch", 0, 2, [7, 9])
run(1, 4096, "torch", 1, 1, [7])
run(2, 4096, "numpy", 1, 1, [7])
run(2, 4096, "torch", 1, 1, [7])
run(2, 4096, "torch", 1, 2, [7])
run(2, 4096, "torch", 0, 2, [7, 9])
