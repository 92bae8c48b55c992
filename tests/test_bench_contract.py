"""The bench line contract (driver + judge read it): the committed line of the last GPU run must carry every required field,
and bench.py must keep the flags the driver passes.  No GPU needed."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest_line():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_bench_v*.json")),
                   key=lambda p: [int(x) for x in re.findall(r"\d+", os.path.basename(p))])
    assert files, "no committed bench line under profiles/"
    with open(files[-1]) as f:
        return json.loads(f.read().strip().splitlines()[-1]), files[-1]


def _latest_detail():
    """the full record of the same run: up to round 5 the line itself, from round 6 on the BENCH_DETAIL line printed before it"""
    d, path = _latest_line()
    with open(path) as f:
        for ln in f.read().strip().splitlines():
            if ln.startswith("BENCH_DETAIL "):
                return json.loads(ln[len("BENCH_DETAIL "):]), path
    return d, path


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_last_line_fits_the_drivers_tail_and_parses():
    """BENCH_r05.json.parsed was null: the line had grown to 24.5 KB and the driver keeps an 8 188-byte tail of stdout.  The last
    line is now bench.compact_line(record); it must stay under 8 000 bytes with every block the judge reads in it - checked on the
    committed line of the last GPU run and on the compact form of the largest full record under profiles/."""
    bench = _bench_module()
    assert bench.MAX_LINE_BYTES <= 8000
    d, path = _latest_line()
    with open(path) as f:
        last = f.read().strip().splitlines()[-1]
    if int(re.findall(r"round(\d+)_", os.path.basename(path))[0]) >= 6:
        assert len(last) < 8000, (path, len(last))
    full, _ = _latest_detail()
    c = bench.compact_line(full)
    line = json.dumps(c)
    assert len(line) < 8000, len(line)
    back = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in back, key
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in back["roofline"], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in back["cpu_baseline"], key
    assert set(back["roofline_labelling"]) >= {"c2", "c3", "c5", "c4"} and "labelling" in back["cpu_baseline"]
    # a record three times the size still yields a parseable line: the optional blocks are shed before the contract keys
    fat = json.loads(json.dumps(full))
    fat["legs"]["api"]["bundled_scenes"].update({f"scene{i}": {"wall_s_median": 1.0, "recorded_s": 1.0, "pad": "x" * 40} for i in range(400)})
    assert len(json.dumps(bench.compact_line(fat))) < 8000


def test_committed_bench_line_has_the_contract_fields():
    d, path = _latest_line()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, (path, key)
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] in ("weak", "strong") and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # achieved = algorithmic bytes per launch / the dominant kernel's average launch duration
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-4 * r["achieved"]    # (6 significant digits in the compact line)
    n, m = d["config"]["points"], d["config"]["hypotheses_per_gpu"]
    assert r["algorithmic_bytes_per_launch"] == n * 5 * 8 + m * 12 * 8 + m * 16 + n * 8      # SURVEY 8(d) + the compound vector
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1
    # value = whole-job throughput of the timed steps
    assert abs(d["value"] - n * m * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def test_round5_fields_labelling_baseline_roofline_and_api_legs():
    """the labelling leg of the CPU baseline, a labelling roofline per config and the drop-in calls' wall times
    travel in the driver-run line"""
    d, path = _latest_detail()
    if int(re.findall(r"round(\d+)_", os.path.basename(path))[0]) < 5:
        return
    lab = d["cpu_baseline"]["labelling"]
    assert lab["kind"] == "port" and lab["cores"] == 1 and lab["cycles_per_s"] > 0 and lab["mincuts_per_s"] > 0
    for key in ("c3", "c5"):
        c = lab["configs"][key]
        assert c["solver"] in ("bk", "dinic") and c["seconds"] > 0 and c["cycles"] >= 1 and c["mincuts"] == c["cycles"] * c["labels"]
        assert c[c["solver"] + "_labels_equal_gpu"] is True          # the CPU solver and the GPU ended on the same labelling
        assert abs(c["cycles_per_s"] - c["cycles"] / c["seconds"]) < 1e-9 * c["cycles_per_s"]
    assert lab["configs"]["c3"]["bk_labels_equal_gpu"] and lab["configs"]["c3"]["dinic_labels_equal_gpu"]
    rl = d["roofline_labelling"]
    for key in ("c2", "c3", "c5", "c4"):
        r = rl[key]
        assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
        leg = d["legs"]["labelling_" + key]
        survey = (leg["sites"] * 12 + leg["arcs"] * 12) * leg["sweeps"] + leg["sites"] * 4 * leg["bfs_levels"]
        steps = leg["sweeps"] + leg["bfs_levels"] + leg["global_relabels"]
        if "algorithmic_bytes_survey_formula" in r:      # round 6: a list sweep is charged by the sites it visited
            assert r["algorithmic_bytes_survey_formula"] == survey
            alg = int((leg["sites"] * 12 + leg["arcs"] * 12) * r["all_site_sweeps"] + r["list_sites_visited"] * (12.0 + 12.0 * leg["arcs"] / leg["sites"])
                      + leg["sites"] * 4 * leg["bfs_levels"])
            assert r["all_site_sweeps"] == leg["sweeps"] - leg["list_sweeps"] and r["list_sweeps"] == leg["list_sweeps"]
            assert r["algorithmic_bytes"] == alg <= survey
            assert r["dependent_steps"] == steps and 0 <= r["steps_inside_persistent_launches"] <= steps
            assert r["launches_per_expansion"] <= steps
            if r["list_sweeps"] > 0:
                assert 0 < r["list_sites_visited"] < r["list_sweeps"] * leg["sites"]
        else:
            alg = survey
            assert r["algorithmic_bytes"] == alg
            assert r["launches_per_expansion"] == steps
        assert abs(r["achieved"] - alg / (leg["expansion_ms"] * 1e-3) / 1e9) <= 1e-9 * max(r["achieved"], 1.0)
    api = d["legs"]["api"]
    for key in ("c1_findLines", "c2_findHomographies", "c3_findTwoViewMotions", "c5_findVanishingPoints", "c4_find6DPoses",
                "c4_find6DPoses_16_objects"):
        assert api[key]["wall_s"] > 0 and api[key]["models"] >= 1
    if "c4_find6DPoses_cap_lifted" in api:      # (from bench v3 on: the 16-object leg runs with scoring_exponent=1 and finds all of them)
        assert api["c4_find6DPoses_16_objects"]["models"] == 16 and api["c4_find6DPoses_16_objects"]["misclassification"] < 0.01
    for scene in ("unionhouse", "unihouse", "oldclassicswing", "breadcube", "cubetoy", "book", "tless"):
        assert api["bundled_scenes"][scene]["wall_s_median"] > 0 and api["bundled_scenes"][scene]["recorded_s"] > 0


def test_bench_py_keeps_the_driver_flags_and_defaults():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert f'"{flag}"' in src
    assert re.search(r'"--gpus", type=int, default=1', src)
