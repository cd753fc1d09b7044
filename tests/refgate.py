"""The north_star parity gate (TEST / BENCH-LEG INFRASTRUCTURE, never product): run the reference's own kernel
(reference include/rwkv/cuda/rwkv.cu:493-593 built unmodified into oracle/_ref/libref.so) and the HIP engine on the
SAME device-resident synthetic tensors, the engine teacher-forced with the reference's greedy ids, and compare the
logits of every step with BASELINE.json's tolerance (parity.py) plus the greedy id of every step.

Used by bench.py (7B, 1024 steps: `ref_kernel_baseline` + `parity_vs_reference_kernel`) and by
tests/test_ref_parity_gpu.py (1B5 and 14B at full depth).  Nothing here reads /root/reference at run time."""
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
