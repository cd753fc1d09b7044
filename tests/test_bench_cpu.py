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
