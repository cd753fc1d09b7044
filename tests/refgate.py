"""The north_star parity gate (TEST / BENCH-LEG INFRASTRUCTURE, never product): run the reference's own kernel
(reference include/rwkv/cuda/rwkv.cu:493-593 built unmodified into oracle/_ref/libref.so) and the HIP engine on the
SAME device-resident synthetic tensors, the engine teacher-forced with the reference's greedy ids, and compare the
logits of every step with BASELINE.json's tolerance (parity.py) plus the greedy id of every step.

Used by bench.py (7B, 1024 steps: `ref_kernel_baseline` + `parity_vs_reference_kernel`; the chunk, long-prompt and 96-stream
gates beside the legs that time those paths) and by tests/test_ref_parity_gpu.py (1B5, 7B and 14B at full depth).  Nothing here reads /root/reference at run time."""
import time

import numpy as np

import oracle_lib
import parity


def ref_model_from_torch(ref, mf, tensors, L, D, maxGPT=1):
    """Hand the reference its 46 tensors in file layout WITHOUT a copy: device pointers of the torch tensors for every
    weight slot, a host copy for EMBED only (the reference keeps the table on the host, rwkv.cu:683-684)."""
    embed_host = tensors[mf.EMBED].detach().cpu().numpy()
    ptrs = []
    for i, t in enumerate(tensors):
        if i == mf.EMBED:
            ptrs.append(embed_host.ctypes.data)
        elif t is None:
            ptrs.append(None)
        else:
            ptrs.append(t.data_ptr())
    rm = ref.from_ptrs(L, D, ptrs, maxGPT)
    rm._keep = (embed_host, tensors)
    return rm


def run_gate(rm, em, mf, prompt, steps, budget_s=None, strict=False, what=""):
    """rm: oracle_lib.RefModel, em: engine.RWKV (resident state).  Feeds `prompt` token by token to both, then `steps`
    greedy steps of the reference (out[0] banned as storygen.cpp:66 does), the engine teacher-forced with the reference's
    ids.  Returns a dict; strict=True asserts parity at every step (the tests), otherwise failures are only counted (bench).
    budget_s bounds the wall time of the reference loop: the number of steps actually run is reported."""
    em.reset_state()
    for s in range(5):
        rm.state(s)[:] = 0.0
    lr = le = None
    for tk in prompt:
        lr = rm.forward([tk])[0]
        le = em.forward(int(tk))[: mf.VOCAB]
    worst, bad_steps, first_div, n_ref_s = 0.0, 0, None, 0.0
    ids_ref, ids_eng = [], []
    tk = parity.argmax_ban0(lr)
    done = 0
    for step in range(steps):
        t0 = time.perf_counter()
        lr = rm.forward([tk])[0]
        n_ref_s += time.perf_counter() - t0
        le = em.forward(int(tk))[: mf.VOCAB]
        d = np.abs(le.astype(np.float64) - lr.astype(np.float64))
        mx = float(np.abs(lr).max()); rms = float(np.sqrt((lr.astype(np.float64) ** 2).mean()))
        rel = float(d.max() / mx)
        ok = bool(np.isfinite(le).all()) and rel <= parity.REL and not bool((d > parity.REL * np.abs(lr) + parity.REL * rms).any())
        worst = max(worst, rel)
        gr, ge = parity.argmax_ban0(lr), parity.argmax_ban0(le)
        ids_ref.append(gr); ids_eng.append(ge)
        if gr != ge and first_div is None:
            first_div = step
        if not ok:
            bad_steps += 1
        if strict:
            parity.check_logits(le, lr, f"{what} step {step}")
            parity.check_argmax(le, lr, f"{what} step {step}")     # identical unless the reference's own top-2 margin is inside the band
        tk = gr
        done = step + 1
        if budget_s is not None and n_ref_s > budget_s:
            break
    return dict(steps=done, max_rel=worst, steps_outside_tolerance=bad_steps, ids_identical=first_div is None,
                first_divergence=first_div, ref_seconds=n_ref_s, ids=ids_ref)


def _rows(le, lr, strict, what):
    """compare [T][V] logits row by row; returns (max_rel, rows outside tolerance, rows whose greedy id differs)"""
    worst, bad, diff = 0.0, 0, 0
    for i in range(lr.shape[0]):
        r = lr[i].astype(np.float64); e = le[i].astype(np.float64)
        d = np.abs(e - r)
        mx = float(np.abs(r).max()); rms = float(np.sqrt((r ** 2).mean()))
        rel = float(d.max() / mx)
        ok = bool(np.isfinite(e).all()) and rel <= parity.REL and not bool((d > parity.REL * np.abs(r) + parity.REL * rms).any())
        worst = max(worst, rel)
        bad += 0 if ok else 1
        diff += 0 if parity.argmax_ban0(lr[i]) == parity.argmax_ban0(le[i]) else 1
        if strict:
            parity.check_logits(le[i], lr[i], f"{what} row {i}")
            parity.check_argmax(le[i], lr[i], f"{what} row {i}")
    return worst, bad, diff


def _state_err(em, rm, n):
    """max |engine - reference| over the first n entries of each of the five state arrays, relative to max(1, max|ref|)"""
    out = {}
    for name, g, which in zip("xy aa bb pp dd".split(), em.state.arrays(), range(5)):
        r = rm.state(which)[:n]
        out[name] = float(np.abs(g[:n] - r).max() / max(1.0, float(np.abs(r).max())))
    return out


def run_chunk_gate(rm, em, mf, engine_mod, prompt, decode_steps=8, strict=False, what="", state_tol=1e-4):
    """BASELINE config 5 at its stated size.  rm: RefModel made with maxGPT >= len(prompt); em: engine.RWKV (resident state,
    maxGPT >= len(prompt)).
    (a) GPT mode: the prompt as ONE multi-token call of the reference (RWKV::forward with T tokens, rwkv.h:339-376; the in-kernel
        token loops rwkv.cu:227,279 -- what RWKV::loadContext drives, rwkv.h:395-413) against the engine's chunk path (mm8_seq on
        the int8 matrix cores): every logits row, the five state arrays, then `decode_steps` greedy single-token steps from that
        state, teacher-forced with the reference's ids.
    (b) PARRALEL mode (rwkv.cu:236-240): one step of len(prompt) independent sequences, two rounds so the per-slot state carries
        over, every logits row and all slots of the five state arrays."""
    T = len(prompt)
    LD = rm.L_ * rm.D
    res = {}
    # ---- (a) GPT chunk ----
    em.reset_state()
    for s in range(5):
        rm.state(s)[:] = 0.0
    t0 = time.perf_counter()
    lr = rm.forward(prompt, oracle_lib.MODE_GPT)
    ref_s = time.perf_counter() - t0
    le = em.forward(prompt, engine_mod.MODE_GPT)[: T * mf.VOCAB].reshape(T, mf.VOCAB).copy()
    worst, bad, diff = _rows(le, lr, strict, f"{what} GPT chunk")
    em.pull_state(1)
    serr = _state_err(em, rm, LD)
    if strict:
        assert max(serr.values()) <= state_tol, serr
    tk = parity.argmax_ban0(lr[T - 1])
    dworst, dbad, ddiff = 0.0, 0, 0
    for step in range(decode_steps):
        l1 = rm.forward([tk])
        e1 = em.forward(int(tk))[: mf.VOCAB].reshape(1, mf.VOCAB)
        w, b, d = _rows(e1, l1, strict, f"{what} decode after chunk, step {step}")
        dworst = max(dworst, w); dbad += b; ddiff += d
        tk = parity.argmax_ban0(l1[0])
    res["gpt_chunk"] = dict(rows=T, max_rel=worst, rows_outside_tolerance=bad, rows_greedy_id_differs=diff, state_max_rel=serr,
                            decode_steps_after=decode_steps, decode_max_rel=dworst, decode_steps_outside_tolerance=dbad,
                            decode_ids_identical=ddiff == 0, ref_chunk_seconds=ref_s)
    # ---- (b) PARRALEL step ----
    res["parralel_step"] = run_streams_gate(rm, em, mf, engine_mod, prompt, rounds=2, strict=strict, what=what, state_tol=state_tol)
    return res


def run_streams_gate(rm, em, mf, engine_mod, first, rounds=2, strict=False, what="", state_tol=1e-4):
    """PARRALEL mode (rwkv.cu:236-240) with T = len(first) independent sequences per step: `rounds` steps of the reference's kernel
    against the engine's batched step, so the per-slot state carries over -- every logits row of every round and all T slots of the
    five state arrays.  T > 64 makes the engine run SEVERAL weight passes per step (a 64-row and a 32-row pass at T = 96) as a
    software pipeline over its streams: the `batched_decode.streams_96` leg of bench.py.  rm / em need maxGPT >= T."""
    T = len(first)
    LD = rm.L_ * rm.D
    em.reset_state()
    for s in range(5):
        rm.state(s)[:] = 0.0
    rng = np.random.default_rng(4321)
    worst, bad, diff = 0.0, 0, 0
    ref_s = 0.0
    for rnd in range(rounds):
        toks = list(first) if rnd == 0 else [int(x) for x in rng.integers(2, mf.VOCAB, T)]
        t0 = time.perf_counter()
        lr = rm.forward(toks, oracle_lib.MODE_PARRALEL)
        ref_s += time.perf_counter() - t0
        le = em.forward(toks, engine_mod.MODE_PARRALEL)[: T * mf.VOCAB].reshape(T, mf.VOCAB).copy()
        w, b, d = _rows(le, lr, strict, f"{what} PARRALEL round {rnd}")
        worst = max(worst, w); bad += b; diff += d
    em.pull_state(T)
    serr = _state_err(em, rm, T * LD)
    if strict:
        assert max(serr.values()) <= state_tol, serr
    em.reset_state()
    return dict(slots=T, rounds=rounds, max_rel=worst, rows_outside_tolerance=bad, rows_greedy_id_differs=diff, state_max_rel=serr,
                ref_seconds=ref_s)


def run_long_prompt_gate(rm, em, mf, engine_mod, tokens, ref_chunk=32, decode_steps=4, strict=False, what="", state_tol=1e-4):
    """A prompt of SEVERAL weight passes handed to the engine in ONE call (RWKV::loadContext with maxContext >= the prompt,
    rwkv.h:395-413: 64-row passes, the three-stream software pipeline, the captured pass graphs -- what bench.py's `long_prompt`
    leg times) against the reference's own kernel fed the same tokens in GPT-mode calls of `ref_chunk` tokens (its in-kernel token
    loops, rwkv.cu:227,279; the state carries from call to call on its side): EVERY logits row, the five state arrays after the
    last token, then `decode_steps` greedy single-token steps from that state.  rm needs maxGPT >= ref_chunk, em maxGPT >= len(tokens)."""
    T = len(tokens)
    LD = rm.L_ * rm.D
    em.reset_state()
    for s in range(5):
        rm.state(s)[:] = 0.0
    le = em.forward(tokens, engine_mod.MODE_GPT)[: T * mf.VOCAB].reshape(T, mf.VOCAB).copy()
    worst, bad, diff, ref_s = 0.0, 0, 0, 0.0
    last = None
    for c0 in range(0, T, ref_chunk):
        part = tokens[c0:c0 + ref_chunk]
        t0 = time.perf_counter()
        lr = rm.forward(part, oracle_lib.MODE_GPT)
        ref_s += time.perf_counter() - t0
        w, b, d = _rows(le[c0:c0 + len(part)], lr, strict, f"{what} prompt rows {c0}..")
        worst = max(worst, w); bad += b; diff += d
        last = lr[len(part) - 1].copy()
    em.pull_state(1)
    serr = _state_err(em, rm, LD)
    if strict:
        assert max(serr.values()) <= state_tol, serr
    tk = parity.argmax_ban0(last)
    dworst, dbad, ddiff = 0.0, 0, 0
    for step in range(decode_steps):
        l1 = rm.forward([tk])
        e1 = em.forward(int(tk))[: mf.VOCAB].reshape(1, mf.VOCAB)
        w, b, d = _rows(e1, l1, strict, f"{what} decode after the prompt, step {step}")
        dworst = max(dworst, w); dbad += b; ddiff += d
        tk = parity.argmax_ban0(l1[0])
    em.reset_state()
    return dict(rows=T, reference_calls=f"{(T + ref_chunk - 1) // ref_chunk} GPT-mode calls of {ref_chunk} tokens", max_rel=worst,
                rows_outside_tolerance=bad, rows_greedy_id_differs=diff, state_max_rel=serr, decode_steps_after=decode_steps,
                decode_max_rel=dworst, decode_steps_outside_tolerance=dbad, decode_ids_identical=ddiff == 0, ref_seconds=ref_s)
