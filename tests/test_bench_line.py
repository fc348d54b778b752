"""bench.py's JSON line must fit the driver's 8 KB tail (round 3 printed 34 KB: 175 AUGX_TIMING strings, BENCH_r03.parsed = null)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def fake_stderr(n_batches):
    lines = ["augx timing: model load                      0.007 s", "augx timing: FASTA read                      0.019 s"]
    lines += ["augx timing:   batch on device 0: 1 pieces, 50001 bases: create + upload 0.000 s, decode 0.014 s, paths 0.000 s, destroy 0.000 s "
              "(device memory free 307.4 of 309.2 GB)"] * n_batches
    lines += ["augx timing: cut finder                      2.902 s", "some other line", "augx timing: genes + GFF                     0.006 s"]
    return "\n".join(lines)


def test_timing_parser_counts_the_batch_lines():
    laps = bench.parse_timing(fake_stderr(200))
    assert laps == {"model load": 0.007, "FASTA read": 0.019, "batches": 200, "cut finder": 2.902, "genes + GFF": 0.006}
    assert len(json.dumps(laps)) < 200


def test_line_is_bounded_and_keeps_the_contract_keys():
    laps = bench.parse_timing(fake_stderr(200))
    out = {"metric": "Mbp DNA decoded/sec (whole node), ab-initio human model", "value": 421.37712345678, "unit": "Mbp/s", "n_gpus": 1, "steps": 20,
           "warmup": 5, "ms_per_step": 237.3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "w" * 120, "note": "n" * 200},
           "roofline": {"bound": "hbm", "achieved": 660.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.0825, "traffic": 6.5e10, "traffic_unit": "t" * 80},
           "cpu_baseline": {"value": 0.149, "unit": "Mbp/s", "cores": 1, "kind": "reference", "sample": "s" * 150}}
    for i in range(12):  # secondary legs with long prose and the laps of an executable run
        out["leg%d" % i] = {"value": 1.0 / 3, "unit": "Mbp/s", "laps_s": dict(laps), "region": "r" * (300 + 10 * i), "workload": "x" * 200}
    line = bench.bounded_line(out)
    assert len(line) < bench.LINE_LIMIT <= 6000
    back = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in back
    assert back["config"] == out["config"] and back["cpu_baseline"]["sample"] == out["cpu_baseline"]["sample"]
    assert back["roofline"]["frac"] == 0.0825 and back["roofline"]["traffic_unit"] == "t" * 80
    assert back["value"] == 421.377
    assert back["leg0"]["laps_s"]["batches"] == 200   # numbers survive, prose goes first
    assert back["metric"] == out["metric"]             # (round 6: the long "metric" string was the first thing the trimming deleted)


def test_a_short_line_is_left_alone():
    out = {"metric": "m", "value": 1.5, "leg": {"region": "r" * 300}}
    assert json.loads(bench.bounded_line(out)) == out
