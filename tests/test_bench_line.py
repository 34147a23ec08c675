"""The bench line the driver parses (bench.py: compact_line / emit): BENCH_r05.json had `parsed: null` because the one
JSON object had grown to 27 KB.  The line builder is run here on a canned full record (the round-5 line as committed under
profiles/) and has to come out small, parseable as the LAST stdout line, and carrying the contract's fields."""
import io
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _canned():
    with open(os.path.join(ROOT, "profiles", "r5z_bench_line.json")) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_compact_line_is_small_and_complete():
    import bench
    line = bench.compact_line(_canned(), "bench_detail.json")
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < bench.LINE_BUDGET < 6000
    back = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in back, k
    assert back["metric"].startswith("frontend+localBA frames/sec")
    assert back["value"] > 0 and back["scaling"] == "weak" and back["vs_baseline"] is None
    assert len(back["config"]["workload"]) <= 300 and "model" not in back["config"]
    r = back["roofline"]
    assert r["bound"] in ("hbm", "mfma", "valu") and r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-4)
    assert {"traffic", "unit", "peak", "kernel", "avg_launch_ms"} <= set(r)
    assert back["cpu_baseline"]["value"] > 0 and back["cpu_baseline"]["cores"] >= 1
    assert back["cpu_baseline"]["kind"] in ("port", "reference") and back["cpu_baseline"]["sample"]
    assert back["roofline_mfma"]["frac"] > 0
    assert back["parity_sample"]["keypoint_bytes_equal"] is True
    assert back["single_stream"]["ms_per_frame"] > 0
    assert back["legs_failed"] == []


def test_emit_prints_the_line_last_and_names_failed_legs(tmp_path, monkeypatch, capsys):
    import bench
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    out = _canned()
    out["rig_batch"] = {"error": "RuntimeError('x')"}
    out["single_stream_rig"]["configs3_4cam_kb8"] = {"error": "boom"}
    buf = io.StringIO()
    buf.write("RCCL version : banner\n")
    bench.emit(out, stream=buf)
    last = buf.getvalue().strip().splitlines()[-1]
    back = json.loads(last)
    assert len(last) < bench.LINE_BUDGET
    assert back["legs_failed"] == ["rig_batch", "single_stream_rig.configs3_4cam_kb8"]
    assert back["detail"] == "bench_detail.json"
    with open(tmp_path / "bench_detail.json") as f:
        assert json.load(f)["single_stream"]["drop_in"]["ms_per_frame"] > 0
    assert "bench_detail: {" in capsys.readouterr().err


def test_line_survives_missing_legs():
    """--no-cpu-baseline / N > 1 runs have no cpu_baseline, parity sample or single stream: still a valid line."""
    import bench
    out = _canned()
    for k in ("cpu_baseline", "parity_sample", "single_stream", "single_stream_rig", "rig_batch"):
        out.pop(k)
    back = json.loads(json.dumps(bench.compact_line(out, None)))
    assert back["roofline"]["frac"] > 0 and "cpu_baseline" not in back
