#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric: tokens/s of single-stream greedy decode of an RWKV-4 uint8
model on MI355X, with the achieved fraction of the HBM-read roofline of the dominant kernel and
the CPU restatement timed beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model 7B] [--no-cpu-baseline]

A "step" is one pass of the hot path: one token through embed+ln0, L x {time-mix, channel-mix},
ln_out + head, and the device-side greedy pick that feeds the next step (storygen's loop with
argmax, examples/storygen/storygen.cpp:63-69).  Weights are synthetic (seeded, real shapes; there
is no network for checkpoints), resident in HBM before the timed region; state never leaves the
device inside it.

N > 1 (launched by torch.distributed.run, one rank per GPU): single-stream decode is a strict
chain of layers, so its only shard is the LAYER PIPELINE: rank s holds layers [l0_s, l1_s) and the
residual vector hops rank -> rank+1 over RCCL send/recv (xGMI); N independent streams are kept in
flight, one per stage, so the aggregate is what scales (weak scaling; one stream alone never gets
faster).  `--parallel replicas` runs N independent full-model replicas instead (no data-path
exchange at all).  Barrier + MAX-reduction of the timing as the contract asks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

USABLE_CPUS = None               # (usable CPUs, cgroup quota): read at the top of main()
HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling 6290
I8_MFMA_PEAK_TOPS = 3944.0      # dense v_mfma_i32_16x16x64_i8 peak (MI355X_MICROARCH.md, MFMA table; no sparsity)


def seq_roofline(L, D, V, rows, seconds):
    """Both rooflines of a pass through the chunk path (mm8_seq on the int8 matrix cores), from what the engine really does:
    every layer matrix is read once per weight pass of up to RWKV_SEQ_ROWS (default 64) rows, the head once per 32-row half;
    every weight byte is multiplied with 3 activation limbs of every row of the pass (2 ops per MAC).  `bound` names the
    roofline that is closer to its peak, i.e. the one that would bind first if the path were perfect."""
    try:      # the engine's own rule (engine.hip rwkv_create): any value <= 32 means 32-row passes
        seq_rows = 64 if int(os.environ.get("RWKV_SEQ_ROWS", "64")) > 32 else 32
    except ValueError:
        seq_rows = 32       # (atoi of a non-number is 0)
    if rows <= 32:
        seq_rows = 32
    passes = (rows + seq_rows - 1) // seq_rows
    halves = (rows + 31) // 32
    wbytes = passes * 13 * L * D * D + halves * V * D
    ops = 2 * 3 * (13 * L * D * D + V * D) * rows
    gbps, tops = wbytes / seconds / 1e9, ops / seconds / 1e12
    fh, fm = gbps / HBM_PEAK_GBPS, tops / I8_MFMA_PEAK_TOPS
    return dict(weight_passes=passes, head_passes=halves, weight_bytes=wbytes, weight_GBps=round(gbps, 1), int8_mfma_TOPS=round(tops, 1),
                roofline=dict(hbm=dict(achieved=round(gbps, 1), peak=HBM_PEAK_GBPS, unit="GB/s", frac=round(fh, 4)),
                              mfma_i8=dict(achieved=round(tops, 1), peak=I8_MFMA_PEAK_TOPS, unit="TOP/s", frac=round(fm, 4)),
                              bound="hbm" if fh >= fm else "mfma",
                              method="weight bytes actually streamed (13 L D^2 per weight pass of <= 64 rows + V D per 32-row half) and "
                                     "2 x 3 limbs x weight bytes x rows int8 operations, over the wall time of the call"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)     # BASELINE: 1024-token greedy continuation
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--model", default=os.environ.get("RWKV_BENCH_MODEL"), help="default: 7B on one GPU (BASELINE config 3, the headline); "
                    "14B when the layers are pipelined over N > 1 GPUs (BASELINE config 4)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the timed tokens of the CPU baseline leg (8 to 32 tokens)")
    ap.add_argument("--ref-steps", type=int, default=1024, help="greedy steps of the reference's own kernel (oracle/_ref/libref.so) "
                    "timed on this GPU and used as the parity gate of the engine (north_star: 1024); 0 = skip")
    ap.add_argument("--ref-seconds", type=float, default=150.0, help="wall-time bound of the reference-kernel leg")
    ap.add_argument("--profile-reps", type=int, default=16)
    ap.add_argument("--long-prompt", type=int, default=512, help="tokens of the one-call prompt of the prefill report (0 = skip)")
    ap.add_argument("--no-long-gate", dest="long_gate", action="store_false", help="skip the reference-kernel gates of the long-prompt and "
                    "96-stream legs (the reference needs ~20 s for the 512 tokens)")
    ap.add_argument("--prefill-chunks", type=int, default=4, help="32-token prompt chunks timed for the prefill report (0 = skip)")
    ap.add_argument("--config2-steps", type=int, default=256, help="greedy steps of the 1B5 leg (BASELINE config 2) reported beside the 7B headline; 0 = skip")
    ap.add_argument("--parallel", choices=["pipeline", "replicas"], default=os.environ.get("RWKV_BENCH_PARALLEL", "pipeline"),
                    help="N > 1: layer pipeline over RCCL send/recv with N streams in flight (default), or N independent replicas")
    args = ap.parse_args()
    if args.model is None:
        args.model = "14B" if args.gpus > 1 and args.parallel == "pipeline" else "7B"
    # the CPU baseline's OpenMP placement must be fixed before the first OpenMP runtime of the process is loaded (torch brings one) --
    # and the CPUs this process may use must be read BEFORE that: a bound OpenMP runtime pins the main thread to its first place
    global USABLE_CPUS
    USABLE_CPUS = usable_cpus()
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # RWKV_BENCH_BACKEND=gloo + RWKV_BENCH_ONE_DEVICE=1: dry run of the N > 1 path on a single-GPU box (all ranks on
        # cuda:0, the hop staged through host memory); the driver's multi-GPU run uses RCCL with one rank per GPU
        backend = os.environ.get("RWKV_BENCH_BACKEND", "nccl")
        if os.environ.get("RWKV_BENCH_ONE_DEVICE") == "1":
            local_rank = 0
        kw = dict(device_id=torch.device("cuda", local_rank)) if backend == "nccl" else {}
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=1800), **kw)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)

    import rwkv_cpp_accelerated_amd as pkg
    from rwkv_cpp_accelerated_amd import engine, modelfile as mf
    pkg.build.build_engine()

    L, D = mf.SHAPES[args.model]
    dev = f"cuda:{local_rank}"
    if world > 1 and args.parallel == "pipeline":
        return bench_pipeline(args, dist, rank, local_rank, world, L, D, dev)
    tensors = mf.synthetic_tensors_torch(L, D, seed=args.seed + rank, device=dev)
    torch.cuda.synchronize()
    m = engine.RWKV(device=local_rank, resident=True)
    t0 = time.time()
    m.loadTensors(L, D, tensors, maxGPT=max(32, args.long_prompt) if args.prefill_chunks > 0 else 1)
    load_s = time.time() - t0

    rng = np.random.default_rng(1)
    prompt = [int(x) for x in rng.integers(2, mf.VOCAB, 32)]     # fixed 32-token prompt

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed: prompt ingestion + W warm-up steps
    for tk in prompt:
        m.forward(tk)
    first = int(np.argmax(m.out[1:mf.VOCAB])) + 1
    if args.warmup > 0:
        ids = m.decode_greedy(first, args.warmup)
        first = int(ids[-1])

    sync_all()
    t0 = time.perf_counter()
    ids = m.decode_greedy(first, args.steps)          # synchronises the engine stream before returning
    sync_all()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    tok_s = args.steps * world / dt
    B_tok = m.bytes_per_token()

    # ---- drop-in mode: host-authoritative state, uploaded/downloaded every token (rwkv.h:353,372) ----
    drop_in = None
    if rank == 0:
        m.resident = False
        m.pull_state(1)
        n_di = min(64, args.steps)
        t0 = time.perf_counter()
        tk = int(ids[-1])
        for _ in range(n_di):
            lg = m.forward(tk)
            tk = int(np.argmax(lg[1:mf.VOCAB])) + 1
        drop_in = dict(tokens_per_s=round(n_di / (time.perf_counter() - t0), 2),
                                    note="host state authoritative: 5xLxD f64 up + down and logits down per token (PCIe-inclusive)")
        m.resident = True

    # ---- per-kernel HIP-event timing on the engine's stream (the stream the kernels are launched on) ----
    # (a) one event pair per launch of an eager replay of the token chain (includes ~2.5 us of event
    #     bracketing per launch); (b) ONE event pair around a batch of all L x reps launches of a class:
    #     the per-launch duration quoted in `roofline`, comparable with rocprofv3's kernel durations.
    prof = m.profile_token(token=int(ids[-1]), reps=args.profile_reps)
    per_launch = {}
    for p in prof:
        n = p["reps"] * p["launches_per_token"]
        us = 1e3 * p["ms_total"] / n if n else 0.0
        per_launch[p["name"]] = dict(us_event_pair=us, bytes=p["bytes_per_launch"], launches_per_token=p["launches_per_token"])
    for p in m.profile_batched(token=int(ids[-1]), reps=max(1, args.profile_reps // 4)):
        d = per_launch[p["name"]]
        d["us"] = p["us"]
        d["gbps"] = (d["bytes"] / (p["us"] * 1e-6) / 1e9) if p["us"] > 0 else 0.0
    # dominant kernel = the class with the most device time per token
    dom = max((k for k in per_launch if per_launch[k]["bytes"] > 0 and k != "first"),
              key=lambda k: per_launch[k]["us"] * per_launch[k]["launches_per_token"])
    roof = dict(bound="hbm", kernel=dom, achieved=round(per_launch[dom]["gbps"], 1), peak=HBM_PEAK_GBPS,
                unit="GB/s", frac=round(per_launch[dom]["gbps"] / HBM_PEAK_GBPS, 4),
                launch_us=round(per_launch[dom]["us"], 3), bytes_per_launch=per_launch[dom]["bytes"],
                traffic=None,
                method="algorithmic uint8 weight bytes of one launch / average launch duration; duration = one hipEvent pair "
                       "around a batch of back-to-back launches of the kernel (all layers x reps) on the engine stream, "
                       "right after the timed region; `traffic` is NOT a counter of this run: it is the committed figure of the "
                       "separate rocprofv3 --pmc FETCH_SIZE pass (x2 gfx950 correction), see traffic_source")
    # `traffic`: HBM read bytes per launch of the dominant kernel from the round's own rocprofv3 --pmc FETCH_SIZE pass
    # (tools/gpu_round.sh writes profiles/<round>/hbm_traffic.json together with the sha256 of the kernel source it profiled);
    # a figure collected for ANOTHER kernels.hip.h / engine.hip is not quoted
    roof.update(traffic_lookup(args.model, dom))
    # the north_star's target is stated per mm8_one shape: every weight-streaming kernel of the token with the reference
    # mm8 calls it absorbs (rwkv.cu:267-311 / :58-142), algorithmic uint8 bytes, launch duration, GB/s, fraction of 8 TB/s
    absorbs = dict(att_kvr_wkv="kernel_mm8_threec K+V+R (3 x D->D)", att_out="mm8_one att_out D->D", ffn_rk="mm8_one ffn_k D->4D + ffn_r D->D",
                   ffn_v="mm8_one<float> ffn_v 4D->D", head="mm8_one head D->V")
    shapes = [dict(kernel=k, reference_calls=absorbs[k], bytes=per_launch[k]["bytes"], us=round(per_launch[k]["us"], 3),
                   GBps=round(per_launch[k]["gbps"], 1), frac_of_8TBps=round(per_launch[k]["gbps"] / HBM_PEAK_GBPS, 4),
                   launches_per_token=per_launch[k]["launches_per_token"]) for k in absorbs if k in per_launch]

    line = dict(
        metric="tokens/sec single-stream RWKV-4 uint8 greedy decode",
        value=round(tok_s, 2), unit="tokens/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
        ms_per_step=round(1e3 * dt / args.steps, 5), higher_is_better=True, scaling="weak",
        vs_baseline=None, dtype="u8 weights x 23-bit fixed-point activations, exact 32-bit integer accumulate (tile form: v_mfma_i32_16x16x64_i8 on signed limbs; row form: v_dot4_u32_u8 on the VALU); f32/f64 epilogues, f64 state", data="synthetic",
        config=dict(workload=f"RWKV-4-Raven-{args.model} uint8 single-stream greedy decode "
                             f"(L={L}, D={D}, V={mf.VOCAB}), 32-token prompt then {args.steps}-token continuation, "
                             "device-resident state",
                    parallelism=("1 GPU" if world == 1 else f"{world} independent replicas, one stream per GPU (replicas only)"),
                    launches_per_token=4 * L + 3, bytes_per_token=B_tok),
        roofline=roof,
        mm8_one_shapes=shapes,
        end_to_end=dict(achieved_GBps=round(B_tok * tok_s / world / 1e9, 1),
                        frac_of_8TBps=round(B_tok * tok_s / world / 1e9 / HBM_PEAK_GBPS, 4)),
        kernels={k: dict(us=round(v["us"], 3), GBps=round(v["gbps"], 1), us_event_pair=round(v["us_event_pair"], 3))
                 for k, v in per_launch.items()},
        load_s=round(load_s, 2),
        hbm_resident_bytes=dict(total=m.resident_bytes(), weight_bytes_one_copy=13 * L * D * D + mf.VOCAB * D,
                                note="device bytes of this context: weights + row-sum tables + embedding + state + scratch.  The matrices of a decode "
                                     "kernel class are resident in the ONE layout its kernel streams: the tile image (csrc/tile.hip.h) for the classes in "
                                     "decode_form.tile, the row form for the others, the head in row form.  The chunk path (max_ctx > 1) multiplies with the "
                                     "16-row tile image of every matrix: the same image at 4096 channels, a second copy elsewhere (DESIGN.md 3)"),
        decode_form=(lambda f: dict(mask=f, tile=[n for b, n in enumerate(("att_kvr_wkv", "att_out", "ffn_rk", "ffn_v")) if f >> b & 1],
                                    row=[n for b, n in enumerate(("att_kvr_wkv", "att_out", "ffn_rk", "ffn_v")) if not f >> b & 1] + ["head"]))(m.decode_form()),
    )

    if drop_in is not None:
        line["drop_in_mode"] = drop_in

    # ---- the reference's own generation loop (storygen: out[0] = -99; typical(out, 0.9, 0.8)) with the DEVICE sampler ----
    if rank == 0:
        n_s = min(256, args.steps)
        m.decode_typical(int(ids[-1]), 8, temp=0.9, tau=0.8, seed=7)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.decode_typical(int(ids[-1]), n_s, temp=0.9, tau=0.8, seed=8)
        dts = time.perf_counter() - t0
        line["sampled_decode"] = dict(tokens_per_s=round(n_s / dts, 2), steps=n_s,
                                      note="typical sampling (temp 0.9, tau 0.8) on the device after every token, no host round trip (csrc/sampler.hip.h)")

    # ---- BASELINE config 5 beside it: 32-token prompt chunks through mm8_seq (int8 MFMA), weights read once per chunk ----
    if rank == 0 and args.prefill_chunks > 0:
        m.forward(prompt, engine.MODE_GPT)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.prefill_chunks):
            m.forward(prompt, engine.MODE_GPT)
        torch.cuda.synchronize()
        dtc = (time.perf_counter() - t0) / args.prefill_chunks
        line["prefill"] = dict(tokens_per_chunk=len(prompt), ms_per_chunk=round(dtc * 1e3, 3), tokens_per_s=round(len(prompt) / dtc, 1),
                               **seq_roofline(L, D, mf.VOCAB, len(prompt), dtc),
                               hbm_traffic=prefill_traffic_lookup(args.model, "chunk32") if len(prompt) == 32 else None,
                               note="GPT-mode chunk (RWKV::loadContext path): v_mfma_i32_16x16x64_i8 over three activation limbs; includes logits for all 32 positions.  "
                                    "hbm_traffic: counter-measured HBM bytes of one weight pass (weights + activation images + the per-K-slice partial values written "
                                    "by the GEMMs and read back by the element-wise kernels) over its weight bytes")
        if args.long_prompt >= 64:
            # a prompt of several chunks handed over in ONE call (RWKV::loadContext with maxContext >= the prompt, rwkv.h:395-413):
            # passes of 64 rows (two 32-row halves that share every weight fragment: weights read once per 64 rows, round 4) as a
            # three-stage software pipeline on three streams (engine.hip rwkv_forward).  Timed without the download of the T x V
            # logits (only the last row matters to loadContext).
            import ctypes as C
            lp = [int(x) for x in np.random.default_rng(11).integers(2, mf.VOCAB, args.long_prompt)]
            arr = (C.c_uint64 * len(lp))(*lp)
            def run_lp():
                rc = engine.lib().rwkv_forward(m._h, arr, len(lp), engine.MODE_GPT)
                if rc != 0:
                    raise RuntimeError(engine.lib().rwkv_last_error().decode())
            run_lp()
            t0 = time.perf_counter()
            for _ in range(2):
                run_lp()
            dtl = (time.perf_counter() - t0) / 2
            line["prefill"]["long_prompt"] = dict(prompt_tokens=len(lp), tokens_per_s=round(len(lp) / dtl, 1), ms=round(dtl * 1e3, 2),
                                                  **seq_roofline(L, D, mf.VOCAB, len(lp), dtl),
                                                  hbm_traffic=prefill_traffic_lookup(args.model, "prompt512") if len(lp) == 512 else None,
                                                  note="one rwkv_forward call: passes of 64 rows (two halves per weight fragment; RWKV_SEQ_ROWS=32: the 32-row schedule, bit-identical "
                                                       "results) as a software pipeline over RWKV_SEQ_STAGES (default 3) streams on the one GPU (stage k on pass i while "
                                                       "stage k - 1 is on pass i + 1); RWKV_SEQ_STAGES=1 gives the one-stream schedule")
        # the same kernels as a batched decode step: 32 independent streams (MODE PARRALEL, state slot per stream)
        m.reset_state()
        m.forward(prompt, engine.MODE_PARRALEL)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.prefill_chunks):
            m.forward(prompt, engine.MODE_PARRALEL)
        torch.cuda.synchronize()
        dtb = (time.perf_counter() - t0) / args.prefill_chunks
        line["batched_decode"] = dict(streams=len(prompt), ms_per_step=round(dtb * 1e3, 3), aggregate_tokens_per_s=round(len(prompt) / dtb, 1),
                                      **seq_roofline(L, D, mf.VOCAB, len(prompt), dtb),
                                      note="MODE PARRALEL: one token of each of 32 independent sequences per step, weights read once per step")
        if args.long_prompt >= 96:
            # 96 streams per step = three 32-row passes, pipelined over the stages like the chunks of a long prompt
            import ctypes as C
            many = [int(x) for x in np.random.default_rng(12).integers(2, mf.VOCAB, 96)]
            arrm = (C.c_uint64 * len(many))(*many)
            def run_many():
                if engine.lib().rwkv_forward(m._h, arrm, len(many), engine.MODE_PARRALEL) != 0:
                    raise RuntimeError(engine.lib().rwkv_last_error().decode())
            m.reset_state()
            run_many()
            t0 = time.perf_counter()
            for _ in range(args.prefill_chunks):
                run_many()
            dtm = (time.perf_counter() - t0) / args.prefill_chunks
            line["batched_decode"]["streams_96"] = dict(ms_per_step=round(dtm * 1e3, 3), aggregate_tokens_per_s=round(len(many) / dtm, 1),
                                                        **seq_roofline(L, D, mf.VOCAB, len(many), dtm),
                                                        note="a 64-row and a 32-row pass per step as a software pipeline over the stages (RWKV_SEQ_STAGES)")

    # ---- BASELINE config 2 beside the headline: RWKV-4-Raven-1B5 single-stream decode on the same GPU ----
    if rank == 0 and args.model == "7B" and args.config2_steps > 0:
        line["config2_1B5"] = small_model_leg(mf, engine, "1B5", args.config2_steps, local_rank, dev)

    # ---- the reference's OWN kernel on this GPU, same tensors, same prompt: parity gate + baseline ----
    if rank == 0 and args.ref_steps > 0:
        line.update(ref_kernel_leg(mf, tensors, L, D, prompt, m, args.ref_steps, args.ref_seconds, B_tok))
        # BASELINE config 5 at its stated size: the 32-token prompt as ONE GPT-mode call of the reference kernel (and one
        # 32-slot PARRALEL step) vs the chunk path that the `prefill` / `batched_decode` legs above timed
        if args.prefill_chunks > 0:
            g = chunk_gate_leg(mf, tensors, L, D, prompt, m, long_prompt=args.long_prompt if args.long_gate else 0)
            if "prefill" in line:
                line["prefill"]["parity_vs_reference_kernel"] = g.get("gpt_chunk", g)
                if "long_prompt" in line["prefill"] and "long_prompt" in g:
                    line["prefill"]["long_prompt"]["parity_vs_reference_kernel"] = g["long_prompt"]
            if "batched_decode" in line:
                line["batched_decode"]["parity_vs_reference_kernel"] = g.get("parralel_step", g)
                if "streams_96" in line["batched_decode"] and "streams_96" in g:
                    line["batched_decode"]["streams_96"]["parity_vs_reference_kernel"] = g["streams_96"]

    # ---- CPU baseline: the oracle (CPU restatement of rwkv.cu:493-593) on this box's host cores ----
    if rank == 0 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(pkg, mf, tensors, L, D, prompt, args.cpu_seconds, engine_model=m)

    m.close()
    bad = parity_failures(line) if rank == 0 else []
    if rank == 0:
        line["parity_gates_failed"] = bad
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if bad:
        # the line above is on record; a number whose parity gate tripped must not pass for a result
        print("[bench] PARITY GATE FAILED: " + "; ".join(bad), file=sys.stderr, flush=True)
        raise SystemExit(4)


def parity_failures(node, path=""):
    """every `*_outside_tolerance` count > 0 (and every state error above its tolerance) anywhere in the line, as 'path = value'"""
    out = []
    if isinstance(node, dict):
        for k, v in node.items():
            p = f"{path}.{k}" if path else k
            if k.endswith("_outside_tolerance") and isinstance(v, (int, float)) and v > 0:
                out.append(f"{p} = {v}")
            elif k == "state_max_rel" and isinstance(v, dict):
                tol = node.get("state_tolerance", 1e-4)
                out += [f"{p}.{a} = {b} > {tol}" for a, b in v.items() if not (b <= tol)]
            elif k == "parity_vs_engine" and isinstance(v, dict) and not (v.get("max_rel_logit_err", 0.0) <= v.get("tolerance", 1e-3)):
                out.append(f"{p}.max_rel_logit_err = {v.get('max_rel_logit_err')}")
            else:
                out += parity_failures(v, p)
    elif isinstance(node, list):
        for i, v in enumerate(node):
            out += parity_failures(v, f"{path}[{i}]")
    return out


def pipeline_line(*, model, L, D, world, steps, warmup, dt, one_steps, dt1, per_stage, layer_ranges, transport, hop, parity, cpu, prefill,
                  native_note, tinfo, resident, per_launch):
    """The JSON line of the N > 1 run from what the ranks measured: a PURE function (no GPU, no torch.distributed), so that a CPU test can
    feed it synthetic per-rank results for N = 2 / 4 / 8 -- the one hardware run must not be lost to a field typo (tests/test_bench_cpu.py).
    dt: seconds of `steps` ticks with `world` streams in flight; dt1: seconds of `one_steps` tokens of ONE stream; per_stage: achieved GB/s of
    every rank's stage; resident: device bytes of every rank's context; per_launch: traffic_lookup() per decode kernel class."""
    from rwkv_cpp_accelerated_amd import modelfile as mf
    B_tok = mf.bytes_per_token(L, D)
    tok_s = world * steps / dt
    worst = min(per_stage)
    # counter-measured HBM read bytes of ONE token through the quoted stage (its layers' four launches + the head on the last stage),
    # from the per-launch FETCH_SIZE figures of the committed PMC pass for this model and these sources; None if there is none
    qs = per_stage.index(worst)
    ql0, ql1 = layer_ranges[qs]
    if all(v.get("traffic") is not None for v in per_launch.values()):
        stage_traffic = (ql1 - ql0) * sum(per_launch[k]["traffic"] for k in ("att_kvr_wkv", "att_out", "ffn_rk", "ffn_v")) + (per_launch["head"]["traffic"] if qs == world - 1 else 0)
        stage_alg = (ql1 - ql0) * 13 * D * D + (mf.VOCAB * D if qs == world - 1 else 0)
        traffic_note = dict(traffic=int(stage_traffic), traffic_unit="HBM read bytes per token through the quoted stage", algorithmic_weight_bytes=int(stage_alg),
                            traffic_source=per_launch["ffn_rk"]["traffic_source"])
    else:
        traffic_note = dict(traffic=None, traffic_source=per_launch["ffn_rk"].get("traffic_source"))
    one_tok_s = one_steps / dt1
    return dict(
        metric=f"tokens/sec RWKV-4 uint8 greedy decode, layers pipelined over {world} GPUs, AGGREGATE of {world} streams in flight (one per stage); "
               "one_stream.tokens_per_s is the single-stream rate on the same pipeline",
        value=round(tok_s, 2), unit="tokens/s",
        n_gpus=world, steps=steps, warmup=warmup, ms_per_step=round(1e3 * dt / steps, 5),
        higher_is_better=True, scaling="weak", vs_baseline=None,
        dtype="u8 weights x 23-bit fixed-point activations, exact 32-bit integer accumulate (tile form: v_mfma_i32_16x16x64_i8 on signed limbs; row form: v_dot4_u32_u8 on the VALU); f32/f64 epilogues, f64 state",
        data="synthetic",
        config=dict(workload=f"RWKV-4-Raven-{model} uint8 greedy decode (L={L}, D={D}), layers pipelined over {world} GPUs, "
                             f"{world} independent streams in flight (one per stage), {steps} tokens per stream",
                    parallelism=f"pp{world}: layer pipeline, {transport} (f64[{D}] between stages, greedy id fed back last->first stage)",
                    layer_ranges=layer_ranges, bytes_per_token=B_tok),
        roofline=dict(bound="hbm", kernel="stage (all decode kernels of a rank's layers)", achieved=worst, peak=HBM_PEAK_GBPS, unit="GB/s",
                      frac=round(worst / HBM_PEAK_GBPS, 4), **traffic_note, quoted_stage=qs, per_stage_GBps=per_stage,
                      method="algorithmic bytes of the stage's layers per token x tokens through the stage / wall time of the timed region; "
                             "the slowest stage is quoted"),
        end_to_end=dict(achieved_GBps=round(B_tok * tok_s / 1e9, 1), frac_of_aggregate_peak=round(B_tok * tok_s / 1e9 / (HBM_PEAK_GBPS * world), 4)),
        per_stream_tokens_per_s=round(steps / dt, 2),
        # north_star: "tokens/sec ... at 1 GPU and -- pipelined -- at 2/4/8 GPUs as absolute numbers and as achieved fraction of the HBM-read roofline":
        # ONE stream through the N stages reads every weight byte of the model once per token whichever GPU holds it, so its roofline is ONE GPU's
        one_stream=dict(tokens_per_s=round(one_tok_s, 2), ms_per_token=round(1e3 * dt1 / one_steps, 5), steps=one_steps,
                        achieved_GBps=round(B_tok * one_tok_s / 1e9, 1), frac_of_8TBps=round(B_tok * one_tok_s / 1e9 / HBM_PEAK_GBPS, 4),
                        roofline_tokens_per_s=round(HBM_PEAK_GBPS * 1e9 / B_tok, 1),
                        note=f"ONE stream in flight through the {world} stages (rwkv_pipe_decode_streams, n_streams = 1): "
                             "t_tok(1 GPU) + (N - 1) hops + the fed-back id per token (SURVEY 8e); N - 1 GPUs idle at any time; "
                             "frac_of_8TBps = fraction of ONE GPU's HBM-read roofline (every weight byte is read once per token, on the GPU that holds it)"),
        hbm_resident_bytes=dict(per_rank=[int(b) for b in resident], total=int(sum(resident)),
                                note="device bytes of every rank's stage context (its layers' matrices in the layout their decode kernels stream, "
                                     "the embedding on rank 0, the head on the last rank, state for 2 N slots, scratch)"),
        hop=hop, parity_vs_single_gpu=parity, cpu_baseline=cpu,
        prefill=prefill, transport_fallback=native_note, transport_info=tinfo)


def bench_pipeline(args, dist, rank, local_rank, world, L, D, dev):
    """N > 1 (BASELINE config 4: RWKV-4-Raven-14B by default): the model's layers are pipelined across the N GPUs (stage s = rank s
    holds layers [l0_s, l1_s), stage 0 the embedding, the last stage the head).  The hop is INSIDE the engine: ncclSend / ncclRecv
    (RCCL over xGMI) of the residual vector (f64[D]) on the engine's own stream, the picked id fed back device to device, no host
    wait per tick (rwkv_pipe_decode).  Three numbers:
      * `value`: N independent greedy streams in flight, one per stage, so every GPU is busy -- a "step" = one token of every
        stream; value = N*K tokens / max-over-ranks time (weak scaling: the aggregate is what scales);
      * `one_stream`: ONE stream through all the stages -- the latency of single-stream decode on the pipeline,
        t_tok(1 GPU) + (N - 1) hops + the fed-back id; it does NOT get faster with N;
      * `hop`: event pairs around the per-tick RCCL group on every rank's stream (min = the hop itself).
    Before anything is timed the LAST rank decodes the same streams on a whole-model context of its own GPU and compares
    (`parity_vs_single_gpu`: picks of every stream identical, last-step logits bit-identical).  torch.distributed only carries
    the 128-byte RCCL id, the barriers and the timing reduction.  RWKV_BENCH_BACKEND=gloo: dry run on one box (Python schedule
    over torch P2P ops; with RWKV_RCCL_LIB=tests/_build/libfake_rccl.so the NATIVE schedule over the shared-memory stand-in)."""
    import faulthandler
    import numpy as np
    import torch
    from rwkv_cpp_accelerated_amd import engine, modelfile as mf, pipeline
    # the engine-side RCCL schedule has run with several ranks only over the shared-memory stand-in (tests/fake_rccl.cpp): a transport
    # that hangs on real xGMI must fail visibly (stack dump + exit) instead of stalling the whole run
    faulthandler.dump_traceback_later(int(os.environ.get("RWKV_BENCH_WATCHDOG_S", "900")), exit=True)
    tensors = mf.synthetic_tensors_torch(L, D, seed=args.seed, device=dev)     # same seed on every rank: one model
    l0, l1 = pipeline.partition_layers(L, world, D)[rank]
    native = (dist.get_backend() == "nccl" or bool(os.environ.get("RWKV_RCCL_LIB"))) and os.environ.get("RWKV_BENCH_NATIVE", "1") == "1"
    stage = pipeline.EngineStage(tensors, L, D, l0, l1, n_slots=2 * world, device=local_rank, prefill=native and args.prefill_chunks > 0)
    lastr = rank == world - 1
    if not (lastr or rank == 0):
        del tensors
        tensors = None
        torch.cuda.empty_cache()
    native_note = None
    if native:
        # the engine-side communicator has never met a second GPU before the driver's own multi-GPU run (a gpurun box has one
        # GPU and RCCL refuses two ranks on one device): if ANY rank fails to join, every rank falls back to the Python
        # schedule over torch.distributed's own RCCL point-to-point ops, and the bench line says so (and the run exits non-zero
        # unless RWKV_BENCH_ALLOW_FALLBACK=1: a number measured on the fallback is not the product's)
        ok = 1
        try:
            pipeline.pipe_connect(stage, dist, rank, world)
        except Exception as e:          # noqa: BLE001 -- reported below
            ok = 0
            native_note = f"rank {rank}: {e}"
        flag = torch.tensor([ok], device=dev, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            native = False
            native_note = native_note or "another rank could not join the engine-side RCCL communicator"
            if rank == 0:
                print(f"[bench] native transport unavailable ({native_note}); using torch.distributed P2P", file=sys.stderr, flush=True)
    # every rank's end of the transport (rwkv_pipe_info: device ordinal, PCI bus id, RCCL version + path, agreed prefill rows) goes into
    # the line: the first run of the engine-side schedule on real xGMI must be diagnosable from its record
    tinfo = [None] * world
    try:
        mine_info = stage.m.pipe_info() if native else dict(rank=rank, transport="torch.distributed P2P")
    except Exception as e:          # noqa: BLE001
        mine_info = dict(rank=rank, error=str(e))
    dist.all_gather_object(tinfo, mine_info)
    rng = np.random.default_rng(1)
    first = [int(x) for x in rng.integers(2, mf.VOCAB, world)]

    def run(n, n_streams=None):
        if native:
            return pipeline.run_pipeline_native(stage, rank, world, first, n, n_streams=n_streams)
        return pipeline.run_pipeline(stage, dist, rank, world, first, n, device=dev, n_streams=n_streams)

    def timed(n, n_streams=None):
        dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(n, n_streams)
        dist.barrier(); torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- parity first: the same streams on ONE GPU (the last rank builds a whole-model context beside its stage) ----
    n_par = int(os.environ.get("RWKV_BENCH_PARITY_STEPS", "12"))
    stage.m.reset_state()
    picks = run(n_par)
    parity = None
    if lastr:
        lg_pipe = stage.m.logits(world).reshape(world, mf.VOCAB).copy()          # row k: the last step of stream k
        try:
            ref = engine.RWKV(device=local_rank, resident=True)
            ref.loadTensors(L, D, tensors, maxGPT=1)
            same_ids, same_logits, worst = True, True, 0.0
            for k in range(world):
                ref.reset_state()
                ids = ref.decode_greedy(first[k], n_par)
                same_ids = same_ids and [int(i) for i in ids] == [int(i) for i in picks[k]]
                # logits of the last step: feed the last-but-one pick (or the first token) once more on a fresh replay
                ref.reset_state()
                tk = first[k]
                for i in range(n_par):
                    lg = ref.forward(tk)[: mf.VOCAB]
                    tk = int(ids[i])
                same_logits = same_logits and bool(np.array_equal(lg, lg_pipe[k]))
                worst = max(worst, float(np.abs(lg.astype(np.float64) - lg_pipe[k]).max() / max(1e-30, float(np.abs(lg).max()))))
            ref.close()
            parity = dict(streams=world, steps=n_par, picks_identical=bool(same_ids), last_step_logits_bit_identical=bool(same_logits),
                          max_rel_logit_err=float(f"{worst:.3e}"),
                          note="the pipeline's streams decoded again on a whole-model context on the last rank's GPU, from the same zero state")
        except Exception as e:          # noqa: BLE001
            parity = dict(skipped=f"whole-model context on the last rank: {e}")
    if lastr and rank != 0:
        del tensors
        tensors = None
        torch.cuda.empty_cache()

    # ---- timed: N streams in flight (the headline), then ONE stream in flight, then the hop ----
    if args.warmup > 0:
        run(max(1, args.warmup // world))
    dt = timed(args.steps)
    one_steps = max(4, min(args.steps, int(os.environ.get("RWKV_BENCH_ONE_STREAM_STEPS", "128"))))
    run(2, 1)
    dt1 = timed(one_steps, 1)
    hop = None
    if native:
        stage.m.pipe_profile(True)
        run(min(64, args.steps))
        hs = stage.m.pipe_hop_stats()
        stage.m.pipe_profile(False)
        mine = torch.tensor([hs["mean_us"], hs["min_us"], hs["max_us"]], device=dev, dtype=torch.float64)
        allh = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allh, mine)
        hop = dict(per_rank_us=[dict(mean=round(float(v[0]), 2), min=round(float(v[1]), 2), max=round(float(v[2]), 2)) for v in allh],
                   note="hipEvent pair on the rank's engine stream around the tick's ncclGroupStart .. ncclGroupEnd {send x | send id | recv x | recv id}, "
                        f"{hs['n']} ticks with {world} streams in flight; min = the hop itself (the peer's data was waiting), mean includes waiting for the peer")
    # per-stage roofline: every token of every stream crosses every stage, so stage s streams its share of the bytes
    # world * K times in dt
    mine = torch.tensor([stage.m.bytes_per_token() * world * args.steps / dt / 1e9], device=dev, dtype=torch.float64)
    per_stage = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(per_stage, mine)
    per_stage = [round(float(v.item()), 1) for v in per_stage]
    par_box = [parity]
    dist.broadcast_object_list(par_box, src=world - 1)
    parity = par_box[0]
    resident = [None] * world
    dist.all_gather_object(resident, int(stage.m.resident_bytes()))
    prefill = None
    if native and args.prefill_chunks > 0:
        n_tok = 64 * max(args.prefill_chunks, 2 * world)
        prompt = [int(x) for x in rng.integers(2, mf.VOCAB, n_tok)]
        pipeline.run_prefill_native(stage, rank, prompt, n_tok)
        dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipeline.run_prefill_native(stage, rank, prompt, n_tok)
        dist.barrier(); torch.cuda.synchronize()
        tp = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce(tp, op=dist.ReduceOp.MAX)
        prefill = dict(prompt_tokens=n_tok, tokens_per_s=round(n_tok / float(tp.item()), 1),
                       note="pipelined RWKV::loadContext: 64-token passes (one weight pass per stage) as micro-batches, stage s on pass t - s (rwkv_pipe_prefill)")
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        import rwkv_cpp_accelerated_amd as pkg
        cpu = cpu_baseline(pkg, mf, tensors, L, D, [first[0]] + [int(x) for x in rng.integers(2, mf.VOCAB, 3)], args.cpu_seconds)
    dist.barrier()
    transport = ("RCCL ncclSend/ncclRecv of the residual vector inside the engine, on its stream" if native and dist.get_backend() == "nccl"
                 else "the engine's native schedule over the shared-memory RCCL stand-in (tests/fake_rccl.cpp; dry run)" if native
                 else "torch.distributed P2P ops (Python schedule)" + ("" if native_note is None else " -- FALLBACK: the engine-side RCCL transport did not come up"))
    out = None
    if rank == 0:
        out = pipeline_line(model=args.model, L=L, D=D, world=world, steps=args.steps, warmup=args.warmup, dt=dt, one_steps=one_steps, dt1=dt1,
                            per_stage=per_stage, layer_ranges=pipeline.partition_layers(L, world, D), transport=transport, hop=hop, parity=parity,
                            cpu=cpu, prefill=prefill, native_note=native_note, tinfo=tinfo, resident=resident,
                            per_launch={k: traffic_lookup(args.model, k) for k in ("att_kvr_wkv", "att_out", "ffn_rk", "ffn_v", "head")})
    # ---- two streams per stage in flight on two communicators: one parity's hop under the other's stage (rwkv_pipe_decode_dual) ----
    # LAST, and under its own watchdog: this schedule has never met real RCCL either; if it hangs, the line measured so far is printed
    # with the leg marked as timed out and the run exits non-zero -- a new leg must not cost the record of the established ones
    dual = None
    if native and os.environ.get("RWKV_BENCH_DUAL", "1") == "1":
        import threading

        def give_up():
            if rank == 0 and out is not None:
                out["two_streams_per_stage"] = dict(error="timed out: the two-communicator schedule did not finish")
                print(json.dumps(out), flush=True)
            os._exit(5)
        timer = threading.Timer(float(os.environ.get("RWKV_BENCH_DUAL_TIMEOUT_S", "240")), give_up)
        timer.daemon = True
        timer.start()
        first2 = [int(x) for x in np.random.default_rng(2).integers(2, mf.VOCAB, 2 * world)]
        try:
            stage.m.reset_state()
            pipeline.run_pipeline_native_dual(stage, rank, world, first2, max(2, args.warmup // world))
            dist.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            pipeline.run_pipeline_native_dual(stage, rank, world, first2, args.steps)
            dist.barrier(); torch.cuda.synchronize()
            td = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
            dist.all_reduce(td, op=dist.ReduceOp.MAX)
            dual = dict(streams=2 * world, tokens_per_s=round(2 * world * args.steps / float(td.item()), 2), ms_per_step=round(1e3 * float(td.item()) / args.steps, 5),
                        note="2 x N streams in flight, the streams of even / odd index on their own communicator and HIP stream, the stage alternating "
                             "between them on the engine's stream: a tick costs max(t_stage, t_hop) instead of their sum (rwkv_pipe_decode_dual)")
        except Exception as e:          # noqa: BLE001
            dual = dict(error=str(e)[:300])
        timer.cancel()
    if rank == 0:
        out["two_streams_per_stage"] = dual
        print(json.dumps(out), flush=True)
    faulthandler.cancel_dump_traceback_later()
    dist.barrier()
    dist.destroy_process_group()
    if native_note is not None and os.environ.get("RWKV_BENCH_ALLOW_FALLBACK") != "1":
        raise SystemExit(3)     # the line above is on record; a fallback-transport number must not pass for the product's


def ref_kernel_leg(mf, tensors, L, D, prompt, engine_model, steps, budget_s, B_tok):
    """rank 0, N=1: the reference's own kernel file (rwkv.cu:493-593, hipcc'd unmodified into oracle/_ref/libref.so --
    test infrastructure, built in the authoring container) decodes `steps` greedy tokens after the same 32-token prompt
    on the same device-resident tensors; the engine is teacher-forced with the reference's ids and its logits are checked
    at every step (north_star: within 1e-3 relative, identical greedy ids).  The reference loop is timed the way its own
    callers run it (RWKV::forward: state upload, kernels, state + logits download, rwkv.h:339-376)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    import refgate
    if not os.path.exists(oracle_lib.REF_SO):
        return dict(ref_kernel_baseline=None, parity_vs_reference_kernel=dict(skipped="oracle/_ref/libref.so not built"))
    ref = oracle_lib.Ref()
    rm = refgate.ref_model_from_torch(ref, mf, tensors, L, D, 1)
    was = engine_model.resident
    engine_model.resident = True
    g = refgate.run_gate(rm, engine_model, mf, prompt, steps, budget_s=budget_s)
    engine_model.resident = was
    tps = g["steps"] / g["ref_seconds"]
    return dict(
        ref_kernel_baseline=dict(tokens_per_s=round(tps, 2), GBps=round(B_tok * tps / 1e9, 1), frac_of_8TBps=round(B_tok * tps / 1e9 / HBM_PEAK_GBPS, 4),
                                 steps=g["steps"], kind="reference",
                                 note="reference include/rwkv/cuda/rwkv.cu built unmodified with hipcc for gfx950, driven through the reference's "
                                      "RWKV::forward (host-authoritative state up/down + logits down per token), same tensors, same prompt"),
        parity_vs_reference_kernel=dict(steps=g["steps"], max_rel=float(f"{g['max_rel']:.3e}"), ids_identical=g["ids_identical"],
                                        first_divergence=g["first_divergence"], steps_outside_tolerance=g["steps_outside_tolerance"],
                                        tolerance=1e-3, note="engine teacher-forced with the reference kernel's greedy ids; logits compared at every step "
                                                             "(max|d| <= 1e-3 max|ref| and |d| <= 1e-3|ref| + 1e-3 rms)"))


def decode_src_digest():
    """sha256 over the sources that decide what a decode launch reads: the kernels (row form, tile form) AND the engine (grid, ring
    geometry, which class runs in which form)"""
    import hashlib
    h = hashlib.sha256()
    for f in ("kernels.hip.h", "tile.hip.h", "engine.hip"):
        h.update(open(os.path.join(ROOT, "rwkv-cpp-accelerated_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def seq_src_digest():
    """sha256 over the sources that decide what a chunk-path pass moves: the mm8_seq kernels and the engine (pass size, GEMM forms, stage split)"""
    import hashlib
    h = hashlib.sha256()
    for f in ("seq.hip.h", "engine.hip"):
        h.update(open(os.path.join(ROOT, "rwkv-cpp-accelerated_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def prefill_traffic_lookup(model, key):
    """counter-measured HBM bytes per weight pass of the chunk path (key "chunk32": one 32-token chunk per call, "prompt512": 64-row passes of
    a 512-token call) from the newest profiles/rNN/prefill_traffic.json whose source digest matches this tree (tools/prefill_traffic.sh)"""
    import glob
    digest = seq_src_digest()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*", "prefill_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        v = d.get(model, {}).get(key)
        if d.get("seq_src_sha256") == digest and v is not None:
            return dict(v, source=f"{os.path.relpath(f, ROOT)} (rocprofv3 --pmc FETCH_SIZE x 2 / WRITE_SIZE, own passes; same seq.hip.h + engine.hip as this run)")
    return dict(source="no PMC pass on record for this seq.hip.h + engine.hip")


def traffic_lookup(model, kernel):
    """HBM bytes per launch of `kernel` from the newest profiles/rNN/hbm_traffic.json whose recorded source digest matches the
    kernels.hip.h + tile.hip.h + engine.hip in the tree (tools/gpu_round.sh records it); {traffic: None, traffic_source: why} otherwise."""
    import glob
    digest = decode_src_digest()
    stale = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*", "hbm_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        rel = os.path.relpath(f, ROOT)
        if d.get("decode_src_sha256") != digest:
            stale.append(rel)
            continue
        v = d.get(model, {}).get(kernel)
        if v is not None:
            return dict(traffic=v, traffic_source=f"{rel} (rocprofv3 --pmc FETCH_SIZE x 2, per launch; same kernels.hip.h + tile.hip.h + engine.hip as this run)")
    return dict(traffic=None, traffic_source="no PMC pass on record for this kernels.hip.h + tile.hip.h + engine.hip" + (f" (other sources: {', '.join(stale[:2])})" if stale else ""))


def small_model_leg(mf, engine, name, steps, device, dev):
    """BASELINE config 2 (RWKV-4-Raven-1B5 uint8 single-stream decode, 1 GPU): same measurement as the headline on a second
    context -- greedy decode after a 32-token prompt, per-kernel batched-event durations, fraction of the HBM roofline."""
    import numpy as np
    import torch
    L, D = mf.SHAPES[name]
    t = mf.synthetic_tensors_torch(L, D, seed=2, device=dev)
    torch.cuda.synchronize()
    m = engine.RWKV(device=device, resident=True)
    m.loadTensors(L, D, t, maxGPT=1)
    for tk in np.random.default_rng(1).integers(2, mf.VOCAB, 32):
        m.forward(int(tk))
    first = int(np.argmax(m.out[1:mf.VOCAB])) + 1
    first = int(m.decode_greedy(first, 16)[-1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ids = m.decode_greedy(first, steps)
    dt = time.perf_counter() - t0
    B = m.bytes_per_token()
    tps = steps / dt
    by = {p["name"]: p["bytes_per_launch"] for p in m.profile_token(token=int(ids[-1]), reps=1)}
    kern = {p["name"]: dict(us=round(p["us"], 3), GBps=round(by[p["name"]] / (p["us"] * 1e-6) / 1e9, 1) if p["us"] > 0 else 0.0)
            for p in m.profile_batched(token=int(ids[-1]), reps=4) if by.get(p["name"], 0) > 0 and p["name"] != "first"}
    m.close()
    del t
    torch.cuda.empty_cache()
    return dict(workload=f"RWKV-4-Raven-{name} uint8 single-stream greedy decode (L={L}, D={D}), {steps} tokens", tokens_per_s=round(tps, 1),
                ms_per_step=round(1e3 * dt / steps, 5), bytes_per_token=B, achieved_GBps=round(B * tps / 1e9, 1),
                frac_of_8TBps=round(B * tps / 1e9 / HBM_PEAK_GBPS, 4), kernels=kern)


def chunk_gate_leg(mf, tensors, L, D, prompt, engine_model, long_prompt=0):
    """rank 0, N=1: full-depth parity of the CHUNK path (mm8_seq on the int8 matrix cores) against the reference's own kernel
    run with T = 32 tokens in one call -- GPT mode (RWKV::loadContext's shape, rwkv.h:339-376,395-413; in-kernel token loops
    rwkv.cu:227,279): all 32 logits rows, the five state arrays, then 8 greedy decode steps from that state; PARRALEL mode
    (rwkv.cu:236-240): two 32-slot steps, all rows, all slots of the state.  tests/refgate.run_chunk_gate.
    long_prompt >= 64: also the legs that run SEVERAL weight passes per call, on the schedule bench.py times (64-row passes, the
    three-stream software pipeline, captured pass graphs): the same `long_prompt` tokens in one rwkv_forward call against the
    reference kernel fed them 32 at a time -- every logits row, the state, 4 decode steps (refgate.run_long_prompt_gate) -- and a
    96-stream PARRALEL step, two rounds, against the reference's 96-slot step (refgate.run_streams_gate)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib
    import refgate
    from rwkv_cpp_accelerated_amd import engine
    if not os.path.exists(oracle_lib.REF_SO):
        return dict(skipped="oracle/_ref/libref.so not built")
    ref = oracle_lib.Ref()
    many = long_prompt >= 96 and engine_model.maxContext >= 96
    rm = refgate.ref_model_from_torch(ref, mf, tensors, L, D, 96 if many else len(prompt))
    was = engine_model.resident
    engine_model.resident = True
    g = refgate.run_chunk_gate(rm, engine_model, mf, engine, prompt, decode_steps=8)
    if long_prompt >= 64 and engine_model.maxContext >= long_prompt:
        lp = [int(x) for x in np.random.default_rng(11).integers(2, mf.VOCAB, long_prompt)]      # the tokens the long_prompt leg timed
        g["long_prompt"] = refgate.run_long_prompt_gate(rm, engine_model, mf, engine, lp, ref_chunk=32, decode_steps=4)
    if many:
        first = [int(x) for x in np.random.default_rng(12).integers(2, mf.VOCAB, 96)]
        g["streams_96"] = refgate.run_streams_gate(rm, engine_model, mf, engine, first, rounds=2)
    engine_model.resident = was
    tol = dict(tolerance=1e-3, state_tolerance=1e-4,
               note="engine chunk path vs reference include/rwkv/cuda/rwkv.cu (built unmodified for gfx950) called with T=32 tokens "
                    "(streams_96: T=96 slots); logits: max|d| <= 1e-3 max|ref| and |d| <= 1e-3|ref| + 1e-3 rms per row; state: max|d| / max(1, max|ref|)")
    for k in g:
        for kk, v in list(g[k].items()):
            if isinstance(v, float):
                g[k][kk] = float(f"{v:.3e}")
            elif isinstance(v, dict):
                g[k][kk] = {a: float(f"{b:.3e}") for a, b in v.items()}
        g[k].update(tol)
    return g


def usable_cpus():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup's CPU quota.  (The GPU boxes show 256 logical CPUs
    and grant 16 CPUs' worth of time -- cpu.max = "1600000 100000": 256 OpenMP threads then take 42 s per 7B token, 64 take 0.2 s;
    profiles/r04/cpu_leg.txt.  That, not "first touch", was the 2x spread of round 3's baseline.)"""
    import math
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(p)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, math.ceil(quota)))
    return n, quota


def cpu_baseline(pkg, mf, tensors, L, D, prompt, budget_s, engine_model=None):
    """rank 0 leg: time the oracle (the CPU restatement of rwkv.cu:493-593; OpenMP over blocks of 64 output columns) on a bounded
    sample of the same workload -- 2 warm-up tokens, then 8-32 timed greedy tokens (as many as fit the budget), each timed on its
    own: min / median / mean are reported, the VALUE is tokens / (sum of the timed tokens).  Threads = the CPUs the process may really
    use (affinity mask capped by the cgroup quota, usable_cpus()); `cores` is that number -- the threads actually used.  OpenMP is
    pinned (OMP_PROC_BIND=close, OMP_PLACES=cores, set in main() before any OpenMP runtime is loaded).  Since the oracle's logits of
    the FULL-SIZE model are at hand, they also check the engine."""
    import glob
    import statistics
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import build_checkers
    build_checkers.build_oracle()
    import oracle_lib
    usable, quota = USABLE_CPUS if USABLE_CPUS else usable_cpus()
    threads = int(os.environ.get("RWKV_BENCH_CPU_THREADS", usable))
    t0 = time.time()
    host = [None if t is None else t.cpu().numpy() for t in tensors]
    copy_s = time.time() - t0
    ora = oracle_lib.Oracle()
    ora.set_threads(threads)      # (torch.distributed.run exports OMP_NUM_THREADS=1 to the ranks of the N > 1 line; the others wait at a barrier)
    om = ora.from_tensors(L, D, host)
    st = om.new_state()
    fed, refs, per = [], [], []
    tk = int(prompt[0])
    n_warm, n = 2, 8
    t_leg = time.perf_counter()
    for i in range(n_warm + 32):
        t0 = time.perf_counter()
        lg = om.forward([tk], st)
        dt1 = time.perf_counter() - t0
        fed.append(tk); refs.append(lg[0].copy())
        tk = int(np.argmax(lg[0][1:])) + 1
        if i == n_warm - 1:
            n = int(max(8, min(32, budget_s // max(dt1, 1e-3))))       # the second warm-up token is the estimate
        if i >= n_warm:
            per.append(dt1)
            if len(per) >= n or (len(per) >= 2 and time.perf_counter() - t_leg > 4 * budget_s):     # (hard bound: a throttled box must not hang the line)
                break
    om.close()
    out = dict(value=round(len(per) / sum(per), 4), unit="tokens/s", cores=ora.num_threads(), kind="port",
               host=dict(logical_cpus=os.cpu_count() or 1, usable_cpus=usable, cgroup_cpu_quota=quota,
                         numa_nodes=len(glob.glob("/sys/devices/system/node/node[0-9]*")) or 1,
                         OMP_PROC_BIND=os.environ.get("OMP_PROC_BIND"), OMP_PLACES=os.environ.get("OMP_PLACES")),
               seconds_per_token=dict(min=round(min(per), 4), median=round(statistics.median(per), 4), mean=round(sum(per) / len(per), 4)),
               sample=f"{len(per)} timed greedy tokens of the same synthetic model after {n_warm} warm-up tokens, each token timed on its own "
                      f"(oracle/rwkv_oracle.c on {ora.num_threads()} OpenMP threads = the CPUs this process may use; weights copied to host in {copy_s:.1f}s)")
    if engine_model is not None:      # full-size parity: same tokens through the engine, logits vs the oracle's
        engine_model.reset_state()
        worst, same = 0.0, True
        for tkn, ref in zip(fed, refs):
            got = engine_model.forward(int(tkn))[: mf.VOCAB]
            worst = max(worst, float(np.abs(got.astype(np.float64) - ref).max() / np.abs(ref).max()))
            same = same and (int(np.argmax(got[1:])) == int(np.argmax(ref[1:])))
        out["parity_vs_engine"] = dict(steps=len(fed), max_rel_logit_err=float(f"{worst:.3e}"), greedy_ids_identical=bool(same),
                                       tolerance=1e-3, note="full-size model, teacher-forced with the oracle's greedy ids")
    return out


if __name__ == "__main__":
    main()
