"""bench.py's output contract, without a GPU: the LAST stdout line is the headline alone - parseable, under 4 KB, carrying
`roofline` and `cpu_baseline` - whatever the `configs` records hold (round 4's line carried them all, grew to 22 KB, and the
driver's record of the round kept only its tail).  The records travel on EARLIER lines and in a details file."""
import io
import json
import os
import subprocess
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def canned_line():
    """A headline of the richest shape bench.py builds (every optional object present, long names)."""
    return {
        "metric": "DP cell-updates/s (GCUPS) on 1M-pair Levenshtein batch", "value": 82383.1, "unit": "GCUPS", "n_gpus": 8, "steps": 200,
        "warmup": 20, "ms_per_step": 0.2084, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32 bit-vectors (u64 results)", "data": "synthetic",
        "config": {"workload": "cfg2: 1024x1024 ASCII len U[96,160], Levenshtein unit [std::mt19937_64]", "pairs_per_gpu": 1048576,
                   "cells_per_gpu": 17151815600, "sharding": "query row blocks, candidates replicated",
                   "entry_point": "szs_levenshtein_distances_u32tape", "generator": "mt19937_64", "stream": "fresh batches alternate",
                   "same_tapes_gcups": 85032.2},
        "roofline": {"bound": "hbm", "achieved": 1614.96, "peak": 8000.0, "unit": "GB/s", "frac": 0.20187, "traffic": 10330183,
                     "kernel_ms": 0.1765, "algorithmic_bytes": 284993536, "launches_per_step": 1, "kernel_gcups": 97193.2, "pmc_stale": False,
                     "pmc_source": "profiles/r05/pmc_configs.json", "kernel": "levenshtein_myers_short_kernel<false>",
                     "valu": {"bound": "int VALU issue, class-weighted", "frac": 0.6727, "achieved_Tlane_ops_per_s": 41.04,
                              "peak_Tlane_ops_per_s": 61.0, "lane_ops_per_cell": 0.4222, "wave_instructions_per_call": 113149755,
                              "full_rate": 513, "half_rate": 208, "lds_busy": 0.3516, "lds_conflict": 0.65, "wait_to_issue": 0.5567, "parked": 0.1612},
                     "peak_measured": 5186.6, "frac_of_measured_peak": 0.311372,
                     "myers_ceiling": {"bound": "int VALU issue, measured (valu_peak.hip)", "achieved_Tcells_per_s_full_width": 108.62,
                                       "peak_Tcells_per_s_full_width": 124.2, "frac": 0.8745, "useful_fraction_of_width": 0.8948}},
        "host_overhead_ms_per_step": 0.0319, "planner": "device, speculated", "results_checksum": 136945828.0,
        "reference_style": {"throughput_gb_s": 1287.1, "efficiency_gops_s": 82383.1, "pairs_per_second": 5031779249.0, "kernel_gcups": 97193.2},
        "same_tapes": {"ms_per_step": 0.2017, "value": 85032.2, "unit": "GCUPS", "calls": 60, "planner_mode": 3},
        "cpu_baseline": {"value": 171.57, "unit": "GCUPS", "cores": 256, "kind": "reference", "tier": "icelake (AVX-512)", "spread": [165.35, 178.43],
                         "sample": "full 1024x1024 batch; median of 20 runs x 11 passes; cells verified equal to the GPU's", "sample_rows": 1024,
                         "sample_columns": 1024, "runs": 20, "passes_per_run": 11, "verified": True,
                         "serial_1_thread": {"value": 19.414, "unit": "GCUPS", "cores": 1, "sample": "4 rows x 64 candidates x 50 passes, serial tier, verified"}},
        "run_seconds": {"total": 93.1, "gpu_legs": 41.0, "cpu_baselines": 52.1},
    }


def canned_records(count=12, padding=2500):
    return [{"config": index, "workload": "w" * 80, "entry_point": "szs_needleman_wunsch_scores_u32tape", "value": 1000.0 + index,
             "unit": "GCUPS", "roofline": {"note": "x" * padding}, "cpu_baseline": {"value": 1.0, "sample": "y" * 200}} for index in range(3, 3 + count)]


def test_the_headline_is_short_parseable_and_complete():
    text = bench.headline(canned_line(), canned_records())
    assert len(text) < 4096 and "\n" not in text
    line = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline", "results_checksum"):
        assert key in line, key
    assert "configs" not in line and line["configs_gcups"]["3"] == 1003.0
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "algorithmic_bytes", "peak_measured", "valu"):
        assert key in line["roofline"], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in line["cpu_baseline"], key
    assert line["config"]["workload"].startswith("cfg2") and line["config"]["entry_point"] and line["config"]["generator"]


def test_the_headline_sheds_optional_objects_before_it_breaks_the_limit():
    fat = canned_line()
    fat["reference_style"]["padding"] = "z" * 3000
    line = json.loads(bench.headline(fat, canned_records(40)))
    assert "reference_style" not in line and "roofline" in line and "cpu_baseline" in line and line["value"] == 82383.1


def test_records_go_first_and_to_the_details_file(tmp_path):
    details = tmp_path / "out" / "bench_configs.json"
    captured = io.StringIO()
    with redirect_stdout(captured):
        bench.emit(canned_line(), canned_records(3), str(details))
    lines = captured.getvalue().splitlines()
    assert len(lines) == 4 and all(json.loads(text)["configs_record"]["config"] in (3, 4, 5) for text in lines[:3])
    last = json.loads(lines[-1])
    assert last["metric"].startswith("DP cell-updates/s") and len(lines[-1]) < 4096
    stored = json.loads(details.read_text())
    assert stored["headline"]["value"] == last["value"] and len(stored["configs"]) == 3
    # what the driver keeps is the tail of stdout: whatever it cuts, the last line survives whole
    assert captured.getvalue()[-8081:].splitlines()[-1] == lines[-1]


def test_more_gpus_than_the_box_has_is_a_parseable_error_not_a_traceback():
    """`python bench.py --gpus 2` with no launcher starts its own ranks; here there is no GPU at all, so it must say so in one
    JSON line and exit 2 (it used to die on `assert world == args.gpus`)."""
    environment = {key: value for key, value in os.environ.items() if key not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    environment["HIP_VISIBLE_DEVICES"] = ""  # also on a GPU box: none visible
    done = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT,
                          capture_output=True, text=True, timeout=600, env=environment)
    assert done.returncode == 2, done.stderr[-2000:]
    line = json.loads(done.stdout.strip().splitlines()[-1])
    assert "error" in line and line["n_gpus"] == 2 and line["visible_gpus"] == 0
    assert "Traceback" not in done.stderr


def test_the_words_record_prices_the_launch_against_its_traffic_and_its_own_instructions():
    """Config 10's `roofline` (bench.words_roofline): HBM-shaped - the results matrix is the algorithmic traffic - beside the counter
    traffic, the staleness flag and the VALU figures of the committed PMC passes, and `issue_floor` from the measured pace of the
    launch's own column; with nothing committed (no counters, no microbenchmark) it still builds."""
    counted = {"traffic": 135023224, "pmc_stale": False, "pmc_source": "profiles/r05/pmc_configs.json", "kernel": "levenshtein_tiny_kernel",
               "valu": {"frac": 0.43, "wave_instructions_per_call": 31206548, "lane_ops_per_cell": 2.65}}
    record = bench.words_roofline(counted, 134305451, 78.1e-6, 1, 5, {"tiny_pure_R16_Tpair_columns": 6.301})
    assert record["bound"] == "hbm" and record["unit"] == "GB/s" and record["algorithmic_bytes"] == 134305451
    assert abs(record["achieved"] - 134305451 / 78.1e-6 / 1e9) < 0.1 and abs(record["frac"] - record["achieved"] / 8000.0) < 1e-3
    assert record["traffic"] == 135023224 and record["pmc_stale"] is False and record["launches_per_step"] == 1 and record["planner_mode"] == 5
    # 31.2 M wavefront-instructions x 2048 pair-columns / 192 instructions / 6.301e12 pair-columns a second = 52.8 us of issue
    assert abs(record["issue_floor"]["floor_ms"] - 0.0528) < 0.0005 and 0.6 < record["issue_floor"]["frac"] < 0.75
    bare = bench.words_roofline({"pmc_stale": None}, 134305451, 78.1e-6, 1, 5, None)
    assert bare["traffic"] is None and "valu" not in bare and "issue_floor" not in bare and bare["kernel"] == "levenshtein_tiny_kernel"
    json.dumps(record), json.dumps(bare)


def test_the_audit_flags_what_round_five_let_through():
    """a fingerprints record at `frac` 1.16 labelled "ESTIMATE: 25 assumed" although a PMC pass of its kernel was
    committed (the key carries the template argument), and a headline naming `levenshtein_myers_short_kernel<false>` as the kernel
    of a stream that only ever launches the fused one.  `bench.audit` names each; a clean line has nothing to say."""
    assert bench.audit(canned_line(), canned_records(3)) == []  # planner "device, speculated" + the plain kernel: consistent
    line = canned_line()
    line["planner"] = "inside the scoring launch"  # ... but the counters joined to it are the plain launch's
    assert any("roofline.kernel" in problem for problem in bench.audit(line))
    line["roofline"]["kernel"] = "levenshtein_myers_short_fused_kernel"
    assert bench.audit(line) == []
    line["roofline"]["valu"]["frac"] = 1.02
    assert any("valu.frac" in problem for problem in bench.audit(line))
    summary = {"cfg11:fingerprint_segments_kernel<true>": {"SQ_INSTS_VALU": 3438668528.0}}
    estimated = {"config": "fingerprints", "roofline": {"counted": "ESTIMATE: 25 assumed", "frac": 1.1623}}
    problems = bench.audit(canned_line(), [estimated], summary)
    assert any("frac" in problem for problem in problems) and any("estimate" in problem for problem in problems)


def test_the_fingerprints_record_is_priced_from_the_committed_counters():
    """`cfg11:fingerprint_segments_kernel<true>` is found by prefix: counted, not estimated, and below 1."""
    summary = {"cfg11:fingerprint_segments_kernel<true>": {"SQ_INSTS_VALU": 3438668528.0}, "_library_sha256": "abc"}
    mixes = {"fingerprint_segments_kernel<true>": {"valu_instructions": 83, "ceiling_Tlane_ops_per_s": 42.92}}
    text_bytes, dimensions = 10485760, 1024
    record = bench.fingerprints_roofline(text_bytes, dimensions, 1024, 5.86e-3, summary, "profiles/r05/pmc_configs.json", mixes, "abc")
    assert record["counted"].startswith("PMC") and record["pmc_stale"] is False and 0.5 < record["frac"] < 1.0
    assert abs(record["lane_ops_per_byte_and_dimension"] - 3438668528.0 * 64 / (text_bytes * dimensions)) < 0.01
    assert record["kernel"] == "fingerprint_segments_kernel<true>" and record["peak_Tlane_ops_per_s"] == 42.92
    # another build of the library: the counters are flagged stale; nothing committed: an estimate, and it says so
    assert bench.fingerprints_roofline(text_bytes, dimensions, 1024, 5.86e-3, summary, "x", mixes, "other")["pmc_stale"] is True
    assert "ESTIMATE" in bench.fingerprints_roofline(text_bytes, dimensions, 1024, 5.86e-3, {}, None, mixes, "abc")["counted"]
    committed, _ = bench._profile_json("pmc_configs.json")
    assert bench._by_prefix(committed, "cfg11:fingerprint_segments_kernel")[1], "the committed passes hold the fingerprints kernel"


def test_a_timed_leg_is_joined_to_the_counters_of_its_own_kernel(monkeypatch):
    """Config 2's run holds two kinds of calls; the headline (fresh batches: the fused launch) takes `cfg2:__call__@fresh`."""
    summary = {"_library_sha256": "abc",
               "cfg2:__call__": {"kernels": {"levenshtein_myers_short_fused_kernel": {"share_of_kernel_time": 0.46},
                                              "levenshtein_myers_short_kernel<false>": {"share_of_kernel_time": 0.54}}},
               "cfg2:__call__@fresh": {"kernels": {"levenshtein_myers_short_fused_kernel": {"share_of_kernel_time": 1.0}},
                                        "hbm_fetch_bytes_raw": 4.9e6, "hbm_write_bytes_raw": 9.4e6, "SQ_INSTS_VALU": 113611086.0},
               "cfg2:__call__@same_tapes": {"kernels": {"levenshtein_myers_short_kernel<false>": {"share_of_kernel_time": 1.0}}}}
    monkeypatch.setattr(bench, "_profile_json", lambda name: (summary, "profiles/rNN/" + name) if name == "pmc_configs.json" else (None, None))
    monkeypatch.setattr(bench, "library_digest", lambda: "abc")
    profile = type("profile", (), {"algorithmic_bytes": 284993536, "launches": 1, "cells": 17151815600})
    fresh = bench.roofline(2, profile, 180e-6, leg="fresh")
    assert fresh["kernel"] == "levenshtein_myers_short_fused_kernel" and fresh["pmc_leg"] == "fresh" and fresh["traffic"] == round(4.9e6 + 9.4e6)
    assert bench.roofline(2, profile, 180e-6, leg="same_tapes")["kernel"] == "levenshtein_myers_short_kernel<false>"
    blend = bench.roofline(2, profile, 180e-6)  # no leg named: the run's average call, as before
    assert blend["pmc_leg"] is None and blend["kernel"] == "levenshtein_myers_short_kernel<false>"


def test_committed_bench_outputs_pass_the_audit():
    """Every bench output committed under profiles/r06/ (the judged command's stdout) is audited again here: no fraction above 1,
    no estimate where counters exist, the headline's kernel is its timed leg's."""
    import glob

    for path in glob.glob(os.path.join(ROOT, "profiles", "r06", "bench_*.jsonl")):
        with open(path) as handle:
            printed = [json.loads(text) for text in handle.read().splitlines() if text.startswith("{")]
        line = printed[-1]
        records = [entry["configs_record"] for entry in printed if "configs_record" in entry]
        assert bench.audit(line, records) == [], path
        assert line.get("audit", []) == [], path
