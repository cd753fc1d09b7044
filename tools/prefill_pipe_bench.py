"""prompt ingestion on ONE GPU as a software pipeline over two co-located stages (layers [0, L/2) and [L/2, L), one stream each):
stage 1 works on chunk c while stage 0 works on chunk c + 1, so every kernel's start-up and tail run under the other stage's
stream.  Compared with the single-context chunk path.  usage: prefill_pipe_bench.py [model] [chunks] [stages]"""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rwkv_cpp_accelerated_amd import engine, modelfile as mf, pipeline
model = sys.argv[1] if len(sys.argv) > 1 else "7B"
nch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
S = int(sys.argv[3]) if len(sys.argv) > 3 else 2
L, D = mf.SHAPES[model]
t = mf.synthetic_tensors_torch(L, D, seed=0)
rng = np.random.default_rng(0)
prompt = [int(x) for x in rng.integers(2, mf.VOCAB, 32 * nch)]
chunks = [prompt[i:i + 32] for i in range(0, len(prompt), 32)]

full = engine.RWKV(resident=True); full.loadTensors(L, D, t, maxGPT=32)
def run_full():
    for c in chunks:
        full.stage_chunk(c, 32, row0=0, buf=0)
    full.sync()
run_full()
t0 = time.perf_counter(); run_full(); dt_full = time.perf_counter() - t0
ref = full.logits(32)[: 32 * mf.VOCAB].reshape(32, mf.VOCAB)[-1].copy()
full.close()

parts = [(L * s // S, L * (s + 1) // S) for s in range(S)]
stages = [pipeline.EngineStage(t, L, D, l0, l1, n_slots=1, prefill=True) for l0, l1 in parts]
del t; torch.cuda.empty_cache()
def run_pipe():
    # plain program order per chunk; the streams and their events do the overlapping
    for ci, c in enumerate(chunks):
        for s in range(S):
            if s > 0:
                stages[s].m.xseq_copy_from(stages[s - 1].m, 32, buf=ci & 1)
            stages[s].m.stage_chunk(c if s == 0 else None, 32, row0=0, buf=ci & 1)
    for st in stages:
        st.m.sync()
run_pipe()
t0 = time.perf_counter(); run_pipe(); dt_pipe = time.perf_counter() - t0
got = stages[-1].m.logits(32)[: 32 * mf.VOCAB].reshape(32, mf.VOCAB)[-1]
same = bool(np.array_equal(got, ref))
print(json.dumps(dict(model=model, chunks=nch, stages=S, single_context_tok_s=round(32 * nch / dt_full, 1), pipelined_tok_s=round(32 * nch / dt_pipe, 1),
                      speedup=round(dt_full / dt_pipe, 3), last_logits_identical=same)))
