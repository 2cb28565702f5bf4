"""bench.py's output contract: the driver keeps the tail of stdout and parses the LAST line, so the headline
must be short, self-contained and last; legs are separate short lines before it.  Round 3's single 24 KB line left
BENCH_r03.json with "parsed": null.  These tests run the formatter on canned records (a real round-3 run and synthetic
worst cases) — no GPU, no library."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def canned():
    with open(os.path.join(ROOT, "profiles", "r03_bench_with_legs.json")) as f:
        full = json.load(f)
    legs = full.pop("legs")
    return full, legs


def test_headline_is_short_and_complete(bench, canned):
    full, legs = canned
    line = bench.dumps_line(bench.compact_headline(full, legs), bench.HEADLINE_MAX_BYTES)
    assert "\n" not in line
    assert len(line) < bench.HEADLINE_MAX_BYTES < 2000 < 4096
    d = json.loads(line)
    # everything the bench contract names
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["config"]["workload"].startswith("direct O(N^2) f32, N=1048576")
    assert "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms"):
        assert k in r, k
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-4)
    assert r["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-5)      # rounding keeps six digits
    assert d["value"] == pytest.approx(full["value"], rel=1e-5)
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert set(d["legs"]) == {l["leg"] for l in legs}
    for ms, frac in d["legs"].values():
        assert ms > 0 and 0 < frac < 1


def test_leg_lines_are_short(bench, canned):
    _, legs = canned
    total = 0
    for leg in legs:
        line = bench.dumps_line(bench.compact_leg(leg), bench.LEG_MAX_BYTES)
        assert len(line) < bench.LEG_MAX_BYTES, leg["leg"]
        d = json.loads(line)
        assert d["leg"] == leg["leg"] and d["value"] > 0
        if "exact" in leg:
            assert d["exact"]["frac"] == pytest.approx(leg["exact"]["roofline"]["frac"], rel=1e-3)
            assert d["fast"]["kernel_ms"] == pytest.approx(leg["fast"]["roofline"]["kernel_ms"], rel=1e-3)
        else:
            assert d["roofline"]["frac"] == pytest.approx(leg["roofline"]["frac"], rel=1e-5)
        total += len(line) + 1
    # seven such legs and the headline fit the ~8 KB of stdout the driver keeps
    assert total / len(legs) * len(bench.LEGS) + bench.HEADLINE_MAX_BYTES < 8000


def test_oversized_strings_are_trimmed_not_emitted(bench, canned):
    full, legs = canned
    fat = json.loads(json.dumps(full))
    fat["config"]["workload"] = "w" * 5000
    fat["config"]["exchange"] = "x" * 5000
    fat["cpu_baseline"]["sample"] = "s" * 5000
    fat["cpu_baseline"].pop("sample_short", None)
    fat["roofline"]["traffic_source"] = "profiles/" + "p" * 3000 + ".json (note)"
    many = [dict(legs[0], leg="leg%03d" % i) for i in range(200)]
    line = bench.dumps_line(bench.compact_headline(fat, many), bench.HEADLINE_MAX_BYTES)
    assert len(line) < bench.HEADLINE_MAX_BYTES
    d = json.loads(line)
    assert d["roofline"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0 and d["value"] > 0


def test_failed_leg_is_reported_briefly(bench, canned):
    full, legs = canned
    bad = {"leg": "config4", "error": "RuntimeError(" + "x" * 4000 + ")"}
    line = bench.dumps_line(bench.compact_leg(bad), bench.LEG_MAX_BYTES)
    assert len(line) < bench.LEG_MAX_BYTES and json.loads(line)["leg"] == "config4"
    d = json.loads(bench.dumps_line(bench.compact_headline(full, [bad]), bench.HEADLINE_MAX_BYTES))
    assert d["legs"]["config4"] == "error"


def test_emit_headline_prints_it_last_and_writes_the_full_record(bench, canned, tmp_path, capsys):
    full, legs = canned
    path = tmp_path / "sub" / "full.json"
    for leg in legs:
        print(bench.dumps_line(bench.compact_leg(leg), bench.LEG_MAX_BYTES))
    bench.emit_headline(full, legs, str(path))
    out = capsys.readouterr().out.strip().split("\n")
    assert len(out) == len(legs) + 1
    last = json.loads(out[-1])
    assert last["metric"].startswith("pair-interactions/sec") and last["roofline"]["frac"] > 0
    assert all("leg" in json.loads(l) for l in out[:-1])
    whole = json.loads(path.read_text())
    assert whole["roofline"]["pairs_per_launch"] == full["roofline"]["pairs_per_launch"] and len(whole["legs"]) == len(legs)


def test_free_weights_leg_has_no_mass_classes(bench):
    import sys
    import numpy as np
    sys.path.insert(0, ROOT)
    from nbody_simulation_amd import scenes
    w = scenes.free_weights(1 << 20, seed=bench.SEED)
    assert w.dtype == np.uint32 and w.min() >= 1 and w.max() <= 100000
    assert len(np.unique(w)) > 1000                  # > 32 distinct values: neither the equal-mass hoist nor the classes apply
    assert np.array_equal(w, scenes.free_weights(1 << 20, seed=bench.SEED))
    assert np.array_equal(w[1000:2000], scenes.free_weights(1000, seed=bench.SEED, start=1000))
    assert "free_masses" in bench.LEGS and "mass_classes" in bench.LEGS
