"""CPU: bench.py's roofline accounting on recorded work counters (no GPU): the line's `frac` must equal achieved / peak and stay <= 1
for the shapes measured in profiles/, cache-resident layer structures are counted at their compulsory HBM bytes, and the dense-query
step is priced against the fp32 vector peak."""
import argparse
import json
import os
import sys

import numpy as np
import pytest
import scipy.sparse as smat

from conftest import REPO

sys.path.insert(0, REPO)
import bench  # noqa: E402


class _FakeClib:
    def __init__(self, stats):
        self._stats = stats

    def predict_stats(self, h, q, beam, pp, topk):
        return self._stats


def _args(config="amazon-670k", scale=1.0, steps=100):
    return argparse.Namespace(topk=10, no_stats=False, steps=steps, config=config, scale=scale, opt=[], include_upload=False)


def _recorded(name):
    j = json.loads(open(os.path.join(REPO, "profiles", name)).read().strip().splitlines()[-1])
    return j


def test_amazon_line_recomputes_from_recorded_work():
    j = _recorded("r02_bench_amazon670k_n1.json")
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == bench.HBM_PEAK_GBPS
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.0 < r["frac"] <= 1.0
    # achieved = algorithmic bytes per launch / average launch duration, as written in the line itself
    assert abs(r["achieved"] - r["alg_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1.0
    # the whole step's algorithmic bytes over the step time stay under the HBM peak
    assert r["step_gbps"] <= bench.HBM_PEAK_GBPS
    # measured fabric traffic (lower bound, FETCH_SIZE x 1) does not exceed the algorithmic bytes
    assert r["traffic"] is not None and r["traffic"] <= r["alg_bytes_per_launch"]
    assert r["l2"]["frac"] <= 1.0
    assert j["cpu_baseline"]["kind"] == "reference" and j["parity"]["scores_bit_identical"]


def test_roofline_function_on_synthetic_counters():
    rows, k, nnz_row = 1000, 10, 50
    X = smat.random(rows, 5000, density=nnz_row / 5000, format="csr", dtype=np.float32, random_state=0)
    depth = 2
    stats = [dict(ref_chunk_bytes=1e9, candidates=rows * 16.0, items=rows * 1.0, probes=float(X.nnz), hit_rows=float(X.nnz) * 0.5,
                  hit_entries=float(X.nnz) * 4, item_cols=rows * 16.0, x_cols=float(X.nnz) * 16),
             dict(ref_chunk_bytes=5e10, candidates=rows * 800.0, items=rows * 10.0, probes=float(X.nnz) * 10, hit_rows=float(X.nnz) * 4,
                  hit_entries=float(X.nnz) * 60, item_cols=rows * 800.0, x_cols=float(X.nnz) * 800)]
    linfo = [dict(lookup=0, bucket_levels=0, dense=1, dense_bytes=2_000_000, device_bytes=3_000_000),
             dict(lookup=2, bucket_levels=0, dense=0, dense_bytes=0, device_bytes=3_000_000_000)]
    prof = [dict(name="k1q_dense", layer=0, ms=0.5, launches=10), dict(name="k0_prolongate", layer=1, ms=0.1, launches=10),
            dict(name="k1_sparse", layer=1, ms=20.0, launches=10), dict(name="k2_topk", layer=1, ms=1.0, launches=10)]
    r = bench.roofline(_FakeClib(stats), None, None, X, prof, linfo, 10, _args(config="unit-test", steps=10), k, rows, 1, 2.2)
    assert r["bound"] == "hbm" and r["kernel"] == "k1_sparse" and r["traffic"] is None
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # no counter file for this configuration: the line says that its frac is the matched-work one, and carries the other two models
    assert r["basis"].startswith("matched-work") and abs(r["frac"] - r["frac_matched"]) < 1e-3 and r["requests"] is None and r["issue"] is None
    assert r["frac_ref_layout"] is not None and abs(r["frac_ref_layout"] - r["alg_bytes_ref_layout"] / 2.2e-3 / 1e9 / r["peak"]) < 0.01
    kern = {(e["name"], e["layer"]): e for e in r["kernels"]}
    # the 3 GB tile structure is larger than the on-chip cache: matched work is the HBM-level figure
    assert kern[("k1_sparse", 1)]["alg_bytes"] == kern[("k1_sparse", 1)]["matched_bytes"]
    # the 2 MB dense level is cache-resident: its HBM-level bytes are the compulsory ones, far below the matched-work bytes
    e0 = kern[("k1q_dense", 0)]
    assert e0["alg_bytes"] < e0["matched_bytes"] and e0["alg_bytes"] >= e0["structure_bytes"]
    # a cache-resident dominant kernel reports the matched-work rate against the L2 peak
    prof2 = [dict(name="k1q_dense", layer=0, ms=50.0, launches=10)]
    r2 = bench.roofline(_FakeClib(stats), None, None, X, prof2, linfo, 10, _args(config="unit-test", steps=10), k, rows, 1, 5.0)
    assert r2["kernel"] == "k1q_dense" and r2["l2"]["bound"] == "l2" and r2["l2"]["peak"] == bench.L2_PEAK_GBPS
    assert abs(r2["l2"]["achieved"] - r2["matched_gbps"]) < 0.2


def test_dense_query_line_is_priced_on_flops():
    Xd = np.ones((64, 32), np.float32)
    stats = [dict(ref_chunk_bytes=1e6, candidates=64 * 16.0, items=64.0, probes=64.0 * 32, hit_rows=64.0 * 32, hit_entries=64.0 * 32 * 16,
                  item_cols=64 * 16.0, x_cols=64.0 * 32 * 16)]
    linfo = [dict(lookup=0, bucket_levels=0, dense=1, dense_bytes=4096, device_bytes=8192)]
    prof = [dict(name="k1g_dense_x", layer=0, ms=1.0, launches=10), dict(name="k2_topk", layer=0, ms=0.1, launches=10)]
    r = bench.roofline(_FakeClib(stats), None, None, Xd, prof, linfo, 10, _args(config="unit-test", steps=10), 10, 64, 1, 0.2)
    assert r["bound"] == "valu" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    assert abs(r["flops_per_launch"] - 2.0 * 64 * 32 * 16) < 1e-6
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3


def test_counter_based_frac_when_the_counter_file_matches(monkeypatch):
    # profiles/pmc_traffic.json holds one counter set per (config, scale, options), each stamped with the hash of the sources it was taken
    # on: for that key ON THOSE SOURCES `frac` is counter bytes / kernel time, the request and issue roofs are attached, and frac_matched /
    # frac_ref_layout stand beside it; on other sources the set is reported as stale and the line falls back to matched work
    tj_all = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json")))
    assert tj_all["version"] == 2 and tj_all["entries"]
    key = sorted(k for k in tj_all["entries"] if "+" not in k)[0]
    tj = tj_all["entries"][key]
    assert key == bench.pmc_key(tj["config"], tj["scale"], []) and len(tj["csrc_sha16"]) >= 8
    fam = "k1_sparse" if "k1_sparse" in tj["kernels"] else sorted(tj["kernels"])[0]
    ent = tj["kernels"][fam]
    rows, k = 1000, 10
    X = smat.random(rows, 5000, density=0.01, format="csr", dtype=np.float32, random_state=0)
    stats = [dict(ref_chunk_bytes=5e10, candidates=rows * 800.0, items=rows * 10.0, probes=float(X.nnz) * 10, hit_rows=float(X.nnz) * 4,
                  hit_entries=float(X.nnz) * 60, item_cols=rows * 800.0, x_cols=float(X.nnz) * 800)]
    linfo = [dict(lookup=2, bucket_levels=0, dense=0, dense_bytes=0, device_bytes=3_000_000_000)]
    ms = 9.0
    name = fam if not fam.startswith("k1q") else "k1q_fused_0_0"
    # the two phases of a pruned layer are ONE launch of the family: their times add up, the kernel's counters per step cover both
    prof = [dict(name=name, layer=0, ms=ms * 10 * 0.75, launches=10), dict(name=name + "_rest", layer=0, ms=ms * 10 * 0.25, launches=10)]
    a1 = _args(config=tj["config"], scale=tj["scale"], steps=10)
    monkeypatch.setattr(bench, "csrc_sha16", lambda: tj["csrc_sha16"])
    r = bench.roofline(_FakeClib(stats), None, None, X, prof, linfo, 10, a1, k, rows, tj["n_gpus"], 9.5)
    assert r["basis"].startswith("pmc") and abs(r["traffic"] - ent["hbm_bytes_per_step"]) < 1.0
    assert abs(r["achieved"] - ent["hbm_bytes_per_step"] / (ms * 1e-3) / 1e9) < 1.0 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-3
    assert r["frac_matched"] is not None and r["frac_ref_layout"] is not None
    if ent.get("fabric_read_req_per_step"):
        assert abs(r["requests"]["fabric_req_per_s_G"] - ent["fabric_read_req_per_step"] / (ms * 1e-3) / 1e9) < 0.1
    if ent.get("valu_insts_per_step"):
        assert 0.0 < r["issue"]["valu_busy_frac"] < 2.0
    # tuning options or the upload mode change what runs: the recorded counters then do not apply
    a2 = _args(config=tj["config"], scale=tj["scale"], steps=10); a2.opt = ["sort_min_tiles=1"]
    assert bench.roofline(_FakeClib(stats), None, None, X, prof, linfo, 10, a2, k, rows, tj["n_gpus"], 9.5)["traffic"] is None
    a3 = _args(config=tj["config"], scale=tj["scale"], steps=10); a3.include_upload = True
    assert bench.roofline(_FakeClib(stats), None, None, X, prof, linfo, 10, a3, k, rows, tj["n_gpus"], 9.5)["traffic"] is None
    # other sources than the ones the counters were taken on: stale, said so in `basis`
    monkeypatch.setattr(bench, "csrc_sha16", lambda: "0" * 16)
    r4 = bench.roofline(_FakeClib(stats), None, None, X, prof, linfo, 10, a1, k, rows, tj["n_gpus"], 9.5)
    assert r4["traffic"] is None and "STALE" in r4["basis"]


def test_csrc_hash_follows_the_sources(tmp_path, monkeypatch):
    h0 = bench.csrc_sha16()
    assert len(h0) == 16 and h0 == bench.csrc_sha16()
    import shutil
    shutil.copytree(os.path.join(REPO, "pecos_amd", "csrc"), tmp_path / "pecos_amd" / "csrc", ignore=shutil.ignore_patterns("build*"))
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    assert bench.csrc_sha16() == h0                                        # build products do not count
    with open(tmp_path / "pecos_amd" / "csrc" / "xrl_k1q.hip", "a") as f:
        f.write("\n// edit\n")
    assert bench.csrc_sha16() != h0
    h1 = bench.csrc_sha16()
    for host_only in ("xrl_tfidf.cpp", "xrl_abi.cpp", "xrl_io.cpp", "xrl_mmap.cpp"):     # host-only sources: no kernel launch of the bench path depends on them
        with open(tmp_path / "pecos_amd" / "csrc" / host_only, "a") as f:
            f.write("\n// edit\n")
        assert bench.csrc_sha16() == h1
    with open(tmp_path / "pecos_amd" / "csrc" / "xrl_predict.cpp", "a") as f:            # the launch logic IS on the kernel path
        f.write("\n// edit\n")
    assert bench.csrc_sha16() != h1


def test_no_committed_line_of_this_round_claims_the_impossible():
    # every committed bench line of the current round must carry a roofline block that is true -- no fraction of a
    # hardware peak above 1, no dominant kernel priced with 0 bytes, counter-based blocks quoting the counters' sources
    import glob
    lines = sorted(glob.glob(os.path.join(REPO, "profiles", "r06_bench_*.json")))
    assert len(lines) >= 6
    for path in lines:
        j = json.loads(open(path).read().strip().splitlines()[-1])
        r = j.get("roofline")
        assert j.get("value_definition"), path
        if r is None:
            continue
        assert 0.0 < r["frac"] <= 1.0, (path, r["frac"])
        assert r["achieved"] > 0.0, path
        if r["bound"] == "hbm":
            assert r["alg_bytes_per_launch"] > 0 or r["traffic"], path
            if r.get("traffic") is not None:
                assert r["basis"].startswith("pmc") and "csrc" in r["basis"], path
        for extra in ("requests", "l2"):
            if r.get(extra):
                assert r[extra]["frac"] <= 1.1, (path, extra, r[extra]["frac"])
        if r["bound"] in ("mfma", "valu"):
            assert r.get("frac_of_reachable", 0.0) <= 1.0, path


def test_round3_amazon_line_is_counter_based_and_self_consistent():
    # the line of the final kernels (exact bound pruning + extraction top-k): `frac` is counter bytes / launch time / peak, the request and
    # issue roofs stay under their ceilings, the matched-work figures are flagged as counting the unpruned candidate set
    j = _recorded("r03_bench_amazon670k_n1.json")
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["basis"].startswith("pmc") and r["kernel"].startswith("k1q")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.0 < r["frac"] <= 1.0
    assert abs(r["achieved"] - r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1.0
    assert 0.0 < r["requests"]["frac"] <= 1.05 and 0.0 < r["issue"]["valu_busy_frac"] <= 1.0 and 0.0 < r["issue"]["salu_issue_frac"] <= 1.0
    # matched work = what the kernels evaluate (the stats pass stages the layers like the timed kernels): a real fraction again;
    # only the reference-layout model (unpruned chunk streaming) is far above 1, and the line says so
    assert 0.0 < r["frac_matched"] <= 1.0 and r["frac_ref_layout"] > 1.0 and "UNPRUNED" in r["pruning"]
    assert sum(w["items"] for w in r["work"]) < 490000 * (1 + 2 + 10 + 10 + 10)
    assert r["avg_launch_ms"] <= j["ms_per_step"]
    assert abs(j["value"] - 490000 * 1e3 / j["ms_per_step"]) / j["value"] < 1e-3
    p = j["parity"]
    assert p["timed_output_identical"] and p["timed_output_scores_bit_identical"] and p["scores_bit_identical"] and p["indices_identical"]
    assert j["cpu_baseline"]["kind"] == "reference" and j["cpu_baseline"]["all_cores"]["cores"] >= j["cpu_baseline"]["cores"]


def _kernel_stats(name):
    import csv
    return list(csv.DictReader(open(os.path.join(REPO, "profiles", name))))


def test_round4_lines_agree_with_the_rocprof_summaries():
    # the contract: `roofline.avg_launch_ms` (hipEvents inside bench.py) must agree with the average duration rocprofv3 --kernel-trace --stats
    # reports for the same kernel; and `frac` must be achieved / peak with achieved = counter bytes of the kernel per launch / that time
    for line, stats, kernel in (("r04_bench_amazon670k_n1.json", "r04_bench_amazon670k_kernel_stats.csv", "k1q_kernel<3, 0, false, true, false, false>"),
                                ("r04_bench_amazon670k_hard_n1.json", "r04_bench_amazon670k_hard_kernel_stats.csv", "k1q_kernel<3, 0, false, true, false, true>")):
        j = _recorded(line)
        r = j["roofline"]
        row = next(x for x in _kernel_stats(stats) if kernel in x["Name"])
        assert abs(float(row["AverageNs"]) * 1e-6 - r["avg_launch_ms"]) / r["avg_launch_ms"] < 0.03, (line, row["AverageNs"], r["avg_launch_ms"])
        assert r["basis"].startswith("pmc") and abs(r["achieved"] - r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1.0
        assert abs(r["frac"] - r["achieved"] / 8000.0) < 1e-3 and 0.3 < r["frac"] < 0.6
        assert j["value_definition"].startswith("DEVICE-RESIDENT") and j["value_host_abi"] < j["value"]
        assert abs(j["value"] - 490000 * 1e3 / j["ms_per_step"]) / j["value"] < 1e-3
        p = j["parity"]
        assert p["timed_output_identical"] and p["scores_bit_identical"] and p["indices_identical"]
        assert j["value"] / j["cpu_baseline"]["value"] > 10.0                      # north_star's target (>= 10x the reference CPU), both models


def test_current_lines_agree_with_the_rocprof_summaries():
    # rounds 5-6: K1Q runs as TWO launches per step (fused narrow levels, then the sorted launch of the last dense-format level): the line's
    # avg_launch_ms of the k1q family = the mean of the two kernels' average durations in the rocprofv3 --kernel-trace --stats summary;
    # `frac` = counter bytes of the family per launch / that time / 8 TB/s.  The default line also carries the extra blocks.
    for line, stats, kernels in (("r06_bench_amazon670k_n1.json", "r06_bench_amazon670k_kernel_stats.csv",
                                  ("k1q_kernel<3, 0, false, true, false, false, false>", "k1q_kernel<3, 0, false, false, false, false, true>")),):
        j = _recorded(line)
        r = j["roofline"]
        rows = _kernel_stats(stats)
        avg = sum(float(next(x for x in rows if k in x["Name"])["AverageNs"]) for k in kernels) * 1e-6 / len(kernels)
        assert r["kernel"].startswith("k1q") and r["basis"].startswith("pmc") and "480bcc7b9a2a34a9" in r["basis"]
        assert len(r["launches_priced"]) == 2 and r["launches_per_step"] == 2.0
        assert abs(avg - r["avg_launch_ms"]) / r["avg_launch_ms"] < 0.03, (avg, r["avg_launch_ms"])
        assert abs(r["achieved"] - r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1.0 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-3
        assert 0.25 < r["frac"] < 0.5 and r["requests"]["frac"] < 1.0
        assert abs(j["value"] - 490000 * 1e3 / j["ms_per_step"]) / j["value"] < 1e-3 and j["value"] / j["cpu_baseline"]["value"] > 10.0
        p = j["parity"]
        assert p["timed_output_identical"] and p["scores_bit_identical"] and p["indices_identical"]
    j = _recorded("r06_bench_amazon670k_n1.json")
    hard, t2l = j["extra"]["hard"], j["extra"]["text_to_labels"]
    assert hard["config"] == "amazon-670k-hard" and hard["parity"]["timed_output_identical"] and 10.0 < hard["ms_per_step"] < 25.0
    assert t2l["labels_identical_to_reference"] and t2l["scores_bit_identical_to_reference"] and t2l["value"] > 10 * t2l["reference"]["value"]
    # the hard workload's own line: the tile-format leaf is now the family with the most GPU time, priced with ITS counters
    h = _recorded("r06_bench_amazon670k_hard_n1.json")
    assert h["roofline"]["kernel"].startswith("k1_sparse") and h["roofline"]["basis"].startswith("pmc") and h["roofline"]["issue"]["valu_busy_frac"] > 0.6
    # BASELINE.json configs[4] at its stated size has a valid line again
    d = _recorded("r06_bench_dense768_full_L3M_N1M_n1.json")
    assert "N=1000000" in d["config"]["workload"] and "L=3000000" in d["config"]["workload"] and d["roofline"]["bound"] == "valu"
    assert 0.0 < d["roofline"]["frac"] < 0.5 and d["parity"]["timed_output_identical"] and d["parity"]["timed_output_scores_bit_identical"]


def test_counter_sets_recompute_from_the_committed_passes():
    # profiles/pmc_traffic.json is DERIVED data: every entry must follow from the per-launch counter rows committed beside it
    # (scripts/pmc_traffic.py: sums per kernel over the last four steps / 4)
    import subprocess
    import tempfile
    tj = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json")))["entries"]
    names = {"amazon-670k@1.0": "amazon670k", "amazon-670k-hard@1.0": "amazon670k_hard", "eurlex-4k@1.0": "eurlex4k", "wiki10-31k@1.0": "wiki10",
             "dense-768@0.25": "dense768_L750k_N250k"}
    assert set(names) <= set(tj)
    for key, n in names.items():
        with tempfile.TemporaryDirectory() as d:
            for k in ("fetch", "write", "l2", "sq"):
                os.symlink(os.path.join(REPO, "profiles", f"r06_pmc_{n}_{k}.csv"), os.path.join(d, f"pmc_{k}.csv"))
            cfg, scale = key.split("@")
            out = os.path.join(d, "e.json")
            subprocess.check_call([sys.executable, os.path.join(REPO, "scripts", "pmc_traffic.py"), d, out, "1.0", cfg, scale], stdout=subprocess.DEVNULL)
            e = json.load(open(out))
        assert e["key"] == key
        for fam, want in tj[key]["kernels"].items():
            got = e["kernels"][fam]
            for f in ("hbm_bytes_per_step", "fabric_read_req_per_step", "valu_insts_per_step", "launches_per_step"):
                assert got[f] == pytest.approx(want[f], rel=1e-9), (key, fam, f)
