"""BASELINE config 5: RWKV-4-7B uint8 batch-32 prompt prefill (mm8_seq on the int8 matrix cores).
Prints one JSON line: chunks/s, tokens/s, achieved GB/s on the weight stream, int8 MFMA utilisation.
usage: python tools/prefill_bench.py [--model 7B] [--chunks 8] [--tokens 32]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rwkv_cpp_accelerated_amd import engine, modelfile as mf

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="7B"); ap.add_argument("--chunks", type=int, default=8); ap.add_argument("--tokens", type=int, default=32)
ap.add_argument("--layers", type=int, default=0)
a = ap.parse_args()
L, D = mf.SHAPES[a.model]
if a.layers: L = a.layers
t = mf.synthetic_tensors_torch(L, D, seed=0)
m = engine.RWKV(resident=True); m.loadTensors(L, D, t, maxGPT=a.tokens)
toks = [int(v) for v in np.random.default_rng(1).integers(2, mf.VOCAB, a.tokens)]
m.forward(toks, engine.MODE_GPT)                      # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.chunks):
    m.forward(toks, engine.MODE_GPT)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.chunks
wbytes = 13 * L * D * D + mf.VOCAB * D
macs = wbytes * a.tokens * 3                          # three limb planes per weight byte and token
print(json.dumps({"metric": "prompt prefill tokens/s (GPT-mode chunk, mm8_seq MFMA path)", "model": a.model, "n_layers": L, "tokens_per_chunk": a.tokens,
                  "ms_per_chunk": dt * 1e3, "chunks_per_s": 1 / dt, "value": a.tokens / dt, "unit": "tokens/s",
                  "weight_GBps": wbytes / dt / 1e9, "weight_frac_of_8TBps": wbytes / dt / 8e12,
                  "int8_mfma_TOPS": 2 * macs / dt / 1e12, "int8_mfma_frac_of_3944": 2 * macs / dt / 3944e12,
                  "vs_token_by_token_decode_speedup_at_500tok_s": (a.tokens / dt) / 500.0}))
m.close()
