"""Layer pipeline across the GPUs of one node (BASELINE.json config 4).

Single-stream decode is a strict chain embed -> L x {time-mix, channel-mix} -> head, so the only
natural shard is BY LAYERS: stage s owns layers [l0_s, l1_s) (and their slices of the five state
arrays), stage 0 the embedding, the last stage the head.  One token of one stream visits the stages
in order; the hop is a point-to-point send/recv of the residual vector x (f64[D], 32-40 KB) over
RCCL/xGMI (torch.distributed, backend "nccl"), and the greedy id goes back from the last stage to
stage 0 (one int64).  One stream alone is not faster than on one GPU (t_tok + (S-1) hops), so the
schedule keeps S independent streams in flight, one per stage (stream k lives on state slot k):
at tick t stage s works on stream (t - s) mod S -- every GPU is busy, aggregate throughput ~ S x.

Two transports.  NATIVE (the product path on a multi-GPU node): the hop lives inside the engine -- ncclSend / ncclRecv on
the engine's own stream between the stage's graph launches, the picked id fed back device to device, no host wait inside
the loop (csrc/engine.hip rwkv_pipe_decode / rwkv_pipe_prefill; `run_pipeline_native`, `run_prefill_native` below only
hand the 128-byte RCCL id around).  PYTHON (`run_pipeline`): the same schedule over torch.distributed P2P ops, with the
stage compute behind a small interface so that it can be tested on CPU (gloo) with the oracle as the backend
(tests/test_pipeline_cpu.py) and on one GPU with EngineStages; it waits on the host every tick.
No reference counterpart: the reference is single-device (no NCCL/MPI call sites).
"""
from __future__ import annotations

import numpy as np

from . import modelfile as mf


def partition_layers(n_layers: int, n_stages: int, n_embed: int = 0, head_weight: float | None = None):
    """Balanced contiguous layer ranges.  The last stage also streams the head (V*D bytes =
    V/(13*D) layers' worth, ~0.9 layers at D=4096), so it gets correspondingly fewer layers."""
    if n_stages < 1 or n_stages > n_layers:
        raise ValueError("need 1 <= n_stages <= n_layers")
    if head_weight is None:
        head_weight = mf.VOCAB / (13.0 * n_embed) if n_embed else 0.0
    total = n_layers + head_weight
    bounds, acc = [0], 0.0
    for s in range(1, n_stages):
        target = total * s / n_stages
        l = int(round(target))
        l = max(bounds[-1] + 1, min(l, n_layers - (n_stages - s)))
        bounds.append(l)
    bounds.append(n_layers)
    return [(bounds[i], bounds[i + 1]) for i in range(n_stages)]


class EngineStage:
    """one pipeline stage on one GPU (the HIP engine restricted to its layer range)"""

    def __init__(self, tensors, n_layers, n_embed, l0, l1, n_slots, device=0, prefill=False):
        import torch
        from . import engine
        self.torch = torch
        self.m = engine.RWKV(device=device, resident=True)
        self.m.set_layer_range(l0, l1)
        self.m.loadTensors(n_layers, n_embed, tensors, maxGPT=max(n_slots, 64) if prefill else n_slots)     # 64: micro-batches of one 64-row weight pass
        self.first, self.last = l0 == 0, l1 == n_layers

        class _X:   # alias the engine's residual buffer as a torch tensor (no copy)
            __cuda_array_interface__ = dict(shape=(n_embed,), typestr="<f8", data=(self.m.x_device_ptr(), False), version=2)
        self.x = torch.as_tensor(_X(), device=f"cuda:{device}")

    def forward(self, token, slot, want_pick):
        return self.m.stage_forward(token, slot, want_pick)


def run_pipeline(stage, dist, rank, world, first_tokens, n_steps, device=None, n_streams=None):
    """Greedy decode of `world` independent streams (stream k starts from first_tokens[k]) for
    n_steps tokens each.  Returns the [world][n_steps] picked ids on the LAST stage (rank world-1),
    None elsewhere.  `stage`: .x (residual tensor handed between stages), .forward(token, slot, want_pick).

    Schedule: global ticks t = 0, 1, ...; stage r works on item j = t - r (stream j % S, step j // S).
    At the START of a tick every rank posts, as ONE batch_isend_irecv group (ncclGroupStart/End under
    RCCL -- a send and a recv to different peers issued separately would deadlock the ring), the send
    of what it produced in the previous tick and the receive of what it needs now:
        r -> r+1 : x      iff stage r+1 has work at t
        S-1 -> 0 : pick   iff stage 0 works on a step >= 1 at t   (S <= t < S*n_steps)"""
    torch = __import__("torch")
    S = world
    if S == 1:
        out = np.zeros((1, n_steps), dtype=np.int64)
        tk = int(first_tokens[0])
        for i in range(n_steps):
            tk = stage.forward(tk, 0, want_pick=True); out[0, i] = tk
        return out
    n_items = S * n_steps
    ns = S if n_streams is None else int(n_streams)           # only the first ns streams' slots of the schedule are filled
    has_work = lambda r, t: 0 <= t - r < n_items and (t - r) % S < ns
    picks = np.zeros((S, n_steps), dtype=np.int64) if rank == S - 1 else None
    # a backend that cannot move device tensors point to point (gloo: the single-GPU test of this very code path)
    # gets the hop staged through host memory; RCCL moves stage.x directly
    host_staging = bool(getattr(stage.x, "is_cuda", False)) and dist.get_backend() == "gloo"
    cdev = "cpu" if host_staging else device
    tok_recv = torch.zeros(1, dtype=torch.int64, device=cdev)
    tok_send = torch.zeros(1, dtype=torch.int64, device=cdev)
    x_send = torch.empty_like(stage.x, device=cdev)
    x_recv = torch.empty_like(stage.x, device=cdev) if host_staging else stage.x
    for tick in range(n_items + S - 1):
        ops = []
        feedback = S <= tick < n_items and tick % S < ns
        if rank < S - 1 and has_work(rank + 1, tick):
            x_send.copy_(stage.x)                                   # stage.x is about to be overwritten by the recv
            ops.append(dist.P2POp(dist.isend, x_send, rank + 1))
        if rank == S - 1 and feedback:
            ops.append(dist.P2POp(dist.isend, tok_send, 0))
        if rank > 0 and has_work(rank, tick):
            ops.append(dist.P2POp(dist.irecv, x_recv, rank - 1))
        if rank == 0 and feedback:
            ops.append(dist.P2POp(dist.irecv, tok_recv, S - 1))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            if host_staging and rank > 0 and has_work(rank, tick):
                stage.x.copy_(x_recv)
            if device is not None and str(device).startswith("cuda"):
                torch.cuda.synchronize()                            # the engine runs on its own stream
        if not has_work(rank, tick):
            continue
        j = tick - rank
        stream, step = j % S, j // S
        token = 0
        if rank == 0:
            token = int(first_tokens[stream]) if step == 0 else int(tok_recv.item())
        pick = stage.forward(token, stream, want_pick=(rank == S - 1))
        if rank == S - 1:
            picks[stream, step] = pick
            tok_send[0] = pick
    return picks


def pipe_connect(stage, dist, rank, world):
    """join the engine-side RCCL communicator: rank 0 makes the id, torch.distributed (any backend) only carries its 128 bytes"""
    from . import engine
    box = [engine.RWKV.pipe_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    stage.m.pipe_init(box[0], rank, world)


def run_pipeline_native(stage, rank, world, first_tokens, n_steps, n_streams=None):
    """greedy decode of `world` streams (or only the first n_streams of them: 1 = one stream through all the stages) with the hop
    inside the engine (pipe_connect first).  [world][n_steps] ids on the last rank."""
    return stage.m.pipe_decode(first_tokens if rank == 0 else None, n_steps, world, last=(rank == world - 1), n_streams=n_streams)


def run_pipeline_native_dual(stage, rank, world, first_tokens, n_steps):
    """greedy decode of 2 * world streams, two per stage in flight: the engine's two-communicator schedule (rwkv_pipe_decode_dual), one
    parity's RCCL hop under the other's stage.  [2 * world][n_steps] ids on the last rank.  The stage needs 2 * world state slots."""
    return stage.m.pipe_decode_dual(first_tokens if rank == 0 else None, n_steps, world, last=(rank == world - 1))


def run_prefill_native(stage, rank, tokens, n_tokens):
    """pipelined prompt ingestion (32-token chunks as micro-batches) with the hop inside the engine"""
    stage.m.pipe_prefill(tokens if rank == 0 else None, n_tokens)
