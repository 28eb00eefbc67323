"""The bench line the round committed (profiles/r03_bench.json = `python bench.py` on MI355X): the keys the driver's contract
and the measurement rules name are there, consistent with each other and with the other files in profiles/ it cites.
(CPU test: it reads the committed evidence, it measures nothing.)"""
import csv
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    return json.loads(open(os.path.join(ROOT, "profiles", "r03_bench.json")).read().strip().splitlines()[-1])


def test_contract_keys_and_arithmetic():
    d = _line()
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"].startswith("integrated rays/sec") and base["metric"].startswith("integrated rays/sec")
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["data"] == "synthetic" and d["unit"] == "rays/s"
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = points per second of the timed steps: 131 072-point scans, ms_per_step each
    assert abs(d["value"] - 131072 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    # every leg's map was compared with the CPU checker, and no stream hand-over timed out in a timed region
    sc = d["self_check"]
    assert sc["map_equals_checker"] and sc["legs_agree"] and sc["pointcloud2_equals_checker"] and sc["server_loop_equals_checker"] and sc["other_configs_ok"]
    assert d["pipeline"]["gate_timeouts"] == 0 and d["pipeline"]["scans_per_walk"] >= 1.0
    assert all(v["digest_ok"] for v in d["other_configs"].values())
    assert d["other_configs"]["C5_lidar8cm_colour"]["fast_path_scans"] > 0, "the 8 cm colour configuration did not take the tiled tree update"


def test_roofline_and_cpu_baseline_objects():
    d = _line()
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # achieved = algorithmic bytes per launch / the kernel's average launch duration
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) / r["achieved"] < 1e-6
    # the rocprofv3 figure comes from the committed stats file whose hash the line carries
    f = os.path.join(ROOT, "profiles", "r03_kernel_stats.csv")
    sha = hashlib.sha256(open(f, "rb").read()).hexdigest()[:16]
    assert sha in r["frac_rocprof_source"]
    rows = {row["kernel"].split("<")[0]: float(row["avg_ns"]) for row in csv.DictReader(open(f)) if row.get("avg_ns")}
    assert abs(r["frac_rocprof"] - r["algorithmic_bytes_per_launch"] / (rows[r["kernel"]] * 1e-9) / 1e9 / r["peak"]) < 1e-6
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert c["unit"] == d["unit"]
