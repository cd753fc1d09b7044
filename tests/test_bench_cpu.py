"""Host logic of bench.py that decides what the line SAYS (no GPU): the pass accounting behind the chunk-path rooflines, the gate
that turns a parity failure anywhere in the line into a non-zero exit, and the source digests that key the committed counter
figures (a figure measured on other sources must read as "none on record", never as this tree's)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_chunk_path_rooflines_count_the_passes_really_made(monkeypatch):
    monkeypatch.delenv("RWKV_SEQ_ROWS", raising=False)
    L, D, V = 32, 4096, 50277
    one = bench.seq_roofline(L, D, V, 32, 3.3e-3)                  # a 32-token chunk: one 32-row pass, the head once
    assert (one["weight_passes"], one["head_passes"]) == (1, 1) and one["weight_bytes"] == 13 * L * D * D + V * D
    lp = bench.seq_roofline(L, D, V, 512, 27e-3)                   # 512 tokens: 8 passes of 64 rows, the head per 32-row half
    assert (lp["weight_passes"], lp["head_passes"]) == (8, 16) and lp["weight_bytes"] == 8 * 13 * L * D * D + 16 * V * D
    assert lp["roofline"]["hbm"]["frac"] == pytest.approx(lp["weight_bytes"] / 27e-3 / 8e12, rel=1e-3)
    assert lp["roofline"]["mfma_i8"]["achieved"] == pytest.approx(2 * 3 * (13 * L * D * D + V * D) * 512 / 27e-3 / 1e12, rel=1e-3)
    ragged = bench.seq_roofline(L, D, V, 96, 7e-3)                 # 96 streams: a 64-row and a 32-row pass, three halves
    assert (ragged["weight_passes"], ragged["head_passes"]) == (2, 3)
    monkeypatch.setenv("RWKV_SEQ_ROWS", "32")                      # the 32-row schedule reads the weights per 32 rows
    assert bench.seq_roofline(L, D, V, 512, 35e-3)["weight_passes"] == 16


def test_a_parity_failure_anywhere_in_the_line_is_found():
    ok = {"parity_vs_reference_kernel": {"steps_outside_tolerance": 0, "state_max_rel": {"aa": 1e-6}, "state_tolerance": 1e-4},
          "prefill": {"long_prompt": {"parity_vs_reference_kernel": {"rows_outside_tolerance": 0, "decode_steps_outside_tolerance": 0}}},
          "cpu_baseline": {"parity_vs_engine": {"max_rel_logit_err": 2.8e-6, "tolerance": 1e-3}}}
    assert bench.parity_failures(ok) == []
    bad = json.loads(json.dumps(ok))
    bad["prefill"]["long_prompt"]["parity_vs_reference_kernel"]["rows_outside_tolerance"] = 3
    bad["parity_vs_reference_kernel"]["state_max_rel"]["aa"] = 2e-4
    bad["cpu_baseline"]["parity_vs_engine"]["max_rel_logit_err"] = float("nan")
    got = bench.parity_failures(bad)
    assert len(got) == 3 and any("rows_outside_tolerance = 3" in g for g in got) and any("state_max_rel.aa" in g for g in got)


def test_counter_figures_are_keyed_on_the_sources_they_were_measured_on(tmp_path, monkeypatch):
    d1, d2 = bench.decode_src_digest(), bench.seq_src_digest()
    assert len(d1) == 64 and len(d2) == 64 and d1 != d2
    # a figure on record for these sources is quoted with its file; one for other sources is not quoted at all
    root = tmp_path / "repo"
    (root / "profiles" / "r98").mkdir(parents=True)
    (root / "profiles" / "r99").mkdir(parents=True)
    csrc = root / "rwkv-cpp-accelerated_amd" / "csrc"
    csrc.mkdir(parents=True)
    for f in ("kernels.hip.h", "tile.hip.h", "engine.hip", "seq.hip.h"):
        (csrc / f).write_text("// " + f)
    monkeypatch.setattr(bench, "ROOT", str(root))
    mine = bench.decode_src_digest()
    json.dump({"7B": {"ffn_rk": 111}, "decode_src_sha256": mine}, open(root / "profiles" / "r98" / "hbm_traffic.json", "w"))
    json.dump({"7B": {"ffn_rk": 999}, "decode_src_sha256": "0" * 64}, open(root / "profiles" / "r99" / "hbm_traffic.json", "w"))
    got = bench.traffic_lookup("7B", "ffn_rk")
    assert got["traffic"] == 111 and "r98" in got["traffic_source"]
    (csrc / "tile.hip.h").write_text("// changed")                 # any of the three decode sources changes the key
    got = bench.traffic_lookup("7B", "ffn_rk")
    assert got["traffic"] is None and "no PMC pass on record" in got["traffic_source"]
    json.dump({"7B": {"chunk32": {"read_over_weights": 1.25}}, "seq_src_sha256": bench.seq_src_digest()}, open(root / "profiles" / "r99" / "prefill_traffic.json", "w"))
    assert bench.prefill_traffic_lookup("7B", "chunk32")["read_over_weights"] == 1.25
    assert "read_over_weights" not in bench.prefill_traffic_lookup("7B", "prompt512")
    (csrc / "seq.hip.h").write_text("// changed")
    assert "read_over_weights" not in bench.prefill_traffic_lookup("7B", "chunk32")


@pytest.mark.parametrize("world", [2, 4, 8])
def test_the_multi_gpu_line_is_assembled_from_per_rank_results(world):
    """bench_pipeline's JSON assembler is a pure function (bench.pipeline_line): synthetic per-rank results for the N the driver runs -- the
    one hardware run of the layer pipeline must not be lost to a field typo.  The line must serialise, carry the contract's fields, quote
    the SLOWEST stage, and state the single-stream rate both as tokens/s and as a fraction of ONE GPU's HBM-read roofline (north_star)."""
    from rwkv_cpp_accelerated_amd import modelfile as mf, pipeline
    L, D = mf.SHAPES["14B"]
    ranges = pipeline.partition_layers(L, world, D)
    assert ranges[0][0] == 0 and ranges[-1][1] == L and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    per_stage = [4100.0 + 10 * r for r in range(world)]
    per_stage[world // 2] = 3900.5                                  # the slowest stage
    hop = dict(per_rank_us=[dict(mean=21.0, min=9.5, max=80.0)] * world, note="synthetic")
    parity = dict(streams=world, steps=12, picks_identical=True, last_step_logits_bit_identical=True, max_rel_logit_err=0.0)
    per_launch = {k: dict(traffic=v, traffic_source="profiles/r99/hbm_traffic.json") for k, v in
                  dict(att_kvr_wkv=80e6, att_out=27e6, ffn_rk=133e6, ffn_v=106e6, head=258e6).items()}
    steps, one_steps, dt, dt1 = 64, 128, 64 * 3.1e-3, 128 * 3.3e-3
    line = bench.pipeline_line(model="14B", L=L, D=D, world=world, steps=steps, warmup=8, dt=dt, one_steps=one_steps, dt1=dt1, per_stage=per_stage,
                               layer_ranges=ranges, transport="RCCL ncclSend/ncclRecv", hop=hop, parity=parity, cpu=dict(value=5.0, unit="tokens/s", cores=16, kind="port", sample="x"),
                               prefill=dict(prompt_tokens=1024, tokens_per_s=30000.0), native_note=None, tinfo=[dict(rank=r) for r in range(world)],
                               resident=[2_000_000_000 + r for r in range(world)], per_launch=per_launch)
    line["two_streams_per_stage"] = None
    back = json.loads(json.dumps(line))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "one_stream", "hop", "parity_vs_single_gpu", "hbm_resident_bytes", "transport_info"):
        assert k in back, k
    B = mf.bytes_per_token(L, D)
    assert back["n_gpus"] == world and back["scaling"] == "weak" and back["vs_baseline"] is None and "model" not in back["config"]
    assert back["value"] == pytest.approx(world * steps / dt, rel=1e-4) and back["ms_per_step"] == pytest.approx(1e3 * dt / steps, rel=1e-4)
    r = back["roofline"]
    assert r["achieved"] == 3900.5 and r["quoted_stage"] == world // 2 and r["frac"] == pytest.approx(3900.5 / 8000.0, abs=1e-4) and r["bound"] == "hbm"
    ql0, ql1 = ranges[world // 2]
    last = world // 2 == world - 1                                  # (N = 2: the quoted stage is the one that holds the head)
    assert r["traffic"] == int((ql1 - ql0) * (80e6 + 27e6 + 133e6 + 106e6) + (258e6 if last else 0))
    assert r["algorithmic_weight_bytes"] == (ql1 - ql0) * 13 * D * D + (mf.VOCAB * D if last else 0)
    one = back["one_stream"]
    assert one["tokens_per_s"] == pytest.approx(1 / 3.3e-3, rel=1e-3) and one["frac_of_8TBps"] == pytest.approx(B / 3.3e-3 / 8e12, rel=1e-3)
    assert one["roofline_tokens_per_s"] == pytest.approx(8e12 / B, rel=1e-3)
    assert back["hbm_resident_bytes"]["per_rank"] == [2_000_000_000 + q for q in range(world)] and back["hbm_resident_bytes"]["total"] == sum(2_000_000_000 + q for q in range(world))
    # without a counter pass on record for these sources the traffic reads null, with the reason
    none = {k: dict(traffic=None, traffic_source="no PMC pass on record") for k in per_launch}
    line2 = bench.pipeline_line(model="14B", L=L, D=D, world=world, steps=steps, warmup=8, dt=dt, one_steps=one_steps, dt1=dt1, per_stage=per_stage,
                                layer_ranges=ranges, transport="x", hop=None, parity=None, cpu=None, prefill=None, native_note="rank 1: boom", tinfo=[], resident=[1] * world, per_launch=none)
    assert line2["roofline"]["traffic"] is None and line2["transport_fallback"] == "rank 1: boom" and json.dumps(line2)
    assert bench.parity_failures(back) == []
