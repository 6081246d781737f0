"""The measured evidence behind the bench lines' `bound` is reproducible from what is committed, and labelled for what it is.

profiles/<tag>_bound_evidence.json (what bench.py reads) must be exactly what tools/bound_evidence.py derives from the committed
counter files and times of the same tag -- nobody edits a verdict by hand --, the projection of the multi-GPU lines must say that
it is a projection, and bench.py must take the evidence only from a file measured on THIS build of the device code."""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _evidence_files():
    return sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bound_evidence.json")))


def test_bound_evidence_is_what_the_tool_derives_from_the_committed_counters(tmp_path):
    files = _evidence_files()
    assert files, "no profiles/*_bound_evidence.json"
    f = files[-1]
    tag = os.path.basename(f)[:-len("_bound_evidence.json")]
    committed = json.load(open(f))
    env = dict(os.environ, NORI_EVIDENCE_SHA=committed["device_source_sha"])      # (the hash is of the build that was measured, not of today's tree)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bound_evidence.py"), os.path.join(ROOT, "profiles"), tag, str(tmp_path)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    again = json.load(open(tmp_path / f"{tag}_bound_evidence.json"))
    assert again == committed
    # what the numbers must look like to mean anything
    for cfg in committed["configs"].values():
        lo, hi = cfg["valu_busy_measured_lo_hi"]
        assert 0.0 < lo <= hi < 2.0                      # not clamped: an upper end above 1 says the additive model overprices
        for v in ("valu", "load", "idle"):
            assert cfg["variants"][v]["trace_ms"] > 0.0
        assert cfg["variants"]["valu"]["added_valu_instr"] > 0 and cfg["variants"]["load"]["added_vmem_rd_instr"] > 0
        # idle cycles add no instruction the counters see (what differs is the persistent waves' timing: who refills when)
        assert abs(cfg["variants"]["idle"]["added_valu_instr"]) < 1e-3 * cfg["base"]["valu_instr"]
        assert abs(cfg["variants"]["idle"]["added_vmem_rd_instr"]) < 1e-2 * cfg["base"]["vmem_rd_instr"]
        assert cfg["verdict"] in ("per-wave latency", "valu + vector-memory issue", "vector-memory issue", "valu")
        assert cfg["verdict_from"]


def test_bench_takes_evidence_of_this_build_only(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    files = _evidence_files()
    assert files
    measured = json.load(open(files[-1]))
    monkeypatch.setattr(bench, "device_source_sha", lambda: measured["device_source_sha"])
    got = bench.bound_evidence("pa4-cbox-path_mis")
    assert got is not None and got[1]["workload"] == "pa4-cbox-path_mis" and "variants" in got[1]
    monkeypatch.setattr(bench, "device_source_sha", lambda: "0" * 16)
    assert bench.bound_evidence("pa4-cbox-path_mis") is None      # a stale figure is worse than none


def test_projected_scaling_says_it_is_a_projection():
    f = os.path.join(ROOT, "profiles", "r6_projected_scaling.json")
    d = json.load(open(f))
    assert "PROJECTION" in d["what"] and "not a measured scaling curve" in d["what"]
    for w in d["workloads"].values():
        rows = w["rows"]
        assert [r["n_gpus"] for r in rows] == [1, 2, 4, 8]
        assert all(r["projected_ms"] >= r["share_ms"] for r in rows)
        assert rows[0]["projected_speedup"] == 1.0 and all(r["projected_speedup"] <= r["n_gpus"] * 1.02 for r in rows)
