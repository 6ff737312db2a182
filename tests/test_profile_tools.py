"""CPU tests of the two tools that feed bench.py's `roofline.traffic` / `kernel_ms_profiled`: tools/summarize_profiles.py turns
rocprofv3 CSVs into profiles/*.json, tools/fingerprint.py decides whether a committed figure still belongs to the kernel
sources bench.py runs from.  Synthetic CSVs in the column layout rocprofv3 writes (profiles/r0*_kernel_stats.csv,
gpurun_out/*/_counter_collection.csv)."""

import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ACC = "void zkmi::accumulate_kernel<zkmi::Bn254G1>(unsigned int const*, unsigned int const*, unsigned int*)"
NTT = "void zkmi::ntt_pass_kernel<zkmi::BnFrParams>(unsigned int const*, unsigned int*, int)"


def _write_csv(path, header, rows):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_ALL)
        w.writerow(header)
        w.writerows(rows)


def _counter_rows(kernel, counter, values):
    return [[i, i, "Agent 4", 1, 100, 100, 1024, 7, kernel, 256, 0, 0, 110, 0, 88, counter, v, 0, 1] for i, v in enumerate(values)]


COUNTER_HEADER = ["Correlation_Id", "Dispatch_Id", "Agent_Id", "Queue_Id", "Process_Id", "Thread_Id", "Grid_Size", "Kernel_Id", "Kernel_Name",
                  "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name",
                  "Counter_Value", "Start_Timestamp", "End_Timestamp"]


def test_summarize_profiles_writes_traffic_kernel_average_and_lds_shares(tmp_path):
    from tools import summarize_profiles as S
    from tools.fingerprint import source_fingerprint
    out_root, prof = str(tmp_path / "gpurun_out"), str(tmp_path / "profiles")
    os.makedirs(prof)
    _write_csv(os.path.join(out_root, "pX_bench_kt", "box", "1_kernel_stats.csv"),
               ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"],
               [[ACC, 25, 26_500_000, 1_060_000.0, 70.0, 1_050_000, 1_080_000, 5000.0],
                ["void zkmi::combine_kernel<zkmi::Bn254G1>(unsigned int const*)", 25, 2_000_000, 80_000.0, 5.0, 1, 2, 0.0]])
    # FETCH_SIZE / WRITE_SIZE are in KiB; the gfx950 correction doubles FETCH_SIZE
    _write_csv(os.path.join(out_root, "pX_bench_fetch", "box", "2_counter_collection.csv"), COUNTER_HEADER,
               _counter_rows(ACC, "FETCH_SIZE", [2_000_000.0, 2_100_000.0]))
    _write_csv(os.path.join(out_root, "pX_bench_write", "box", "3_counter_collection.csv"), COUNTER_HEADER,
               _counter_rows(ACC, "WRITE_SIZE", [60_000.0, 60_000.0]))
    # LDS pass over one transform (three launches): conflicts in the first and last pass only
    lds_rows = []
    for name, per_pass in (("SQ_LDS_BANK_CONFLICT", [300.0, 0.0, 300.0]), ("SQ_LDS_IDX_ACTIVE", [800.0, 400.0, 800.0]),
                           ("SQ_WAVE_CYCLES", [10_000.0, 8_000.0, 10_000.0]), ("SQ_WAIT_INST_LDS", [400.0, 200.0, 500.0])):
        lds_rows += _counter_rows(NTT, name, per_pass * 2)
    _write_csv(os.path.join(out_root, "pX_ntt_lds", "box", "4_counter_collection.csv"), COUNTER_HEADER, lds_rows)

    S.main("rXX", "pX", out_root=out_root, prof=prof)

    with open(os.path.join(prof, "msm_traffic.json")) as f:
        t = json.load(f)
    assert t["hbm_bytes"] == int((2 * 2_050_000.0 + 60_000.0) * 1024)
    assert t["kernel"].startswith("zkmi::accumulate_kernel<zkmi::Bn254G1>")
    assert t["kernel_avg_ms_profiled"] == 1.06 and t["kernel_launches_profiled"] == 25
    assert t["source_fingerprint"] == source_fingerprint("msm")
    assert os.path.exists(os.path.join(prof, "rXX_bench_kernel_stats.csv"))
    with open(os.path.join(prof, "rXX_bench_pmc_summary.json")) as f:
        summary = json.load(f)
    (row,) = summary.values()
    assert row["launches_sampled"] == 2 and row["FETCH_SIZE_KiB_raw"] == 2_050_000.0
    with open(os.path.join(prof, "rXX_ntt_pmc_lds.json")) as f:
        lds = json.load(f)
    passes = lds["kernels"]["zkmi::ntt_pass_kernel<zkmi::BnFrParams>"]["pass_0_1_2_of_a_2^22_transform"]
    assert [p["bank_conflict_share_of_lds_active_cycles"] for p in passes] == [0.375, 0.0, 0.375]
    assert passes[2]["wave_cycles_waiting_on_lds_share"] == 0.05


def test_fingerprint_gates_the_committed_figures(tmp_path, monkeypatch):
    from tools import fingerprint as F
    # a private tree: two "kernel sources" and a profiles/ directory
    root = tmp_path
    (root / "profiles").mkdir()
    (root / "a.h").write_text("kernel A\n")
    (root / "b.h").write_text("kernel B\n")
    monkeypatch.setattr(F, "ROOT", str(root))
    monkeypatch.setattr(F, "SOURCES", {"msm": ("a.h", "b.h")})
    fp = F.source_fingerprint("msm")
    assert len(fp) == 16 and fp == F.source_fingerprint("msm")
    (root / "profiles" / "t.json").write_text(json.dumps({"hbm_bytes": 123, "kernel_avg_ms_profiled": 1.07, "source_fingerprint": fp, "source": "s"}))
    assert F.load_traffic("t.json", "msm") == (123, "s")
    assert F.load_profiled("t.json", "msm", "kernel_avg_ms_profiled") == 1.07
    # any change of a source (content, or the same bytes moved between files) retires the figure, with the reason
    (root / "b.h").write_text("kernel B'\n")
    value, why = F.load_traffic("t.json", "msm")
    assert value is None and "other kernel sources" in why and fp in why
    assert F.load_profiled("t.json", "msm", "kernel_avg_ms_profiled") is None
    (root / "a.h").write_text("kernel A\nkernel B\n")
    (root / "b.h").write_text("")
    assert F.source_fingerprint("msm") != fp
    # absent / unreadable files are not figures
    assert F.load_traffic("nope.json", "msm") == (None, None)
    (root / "profiles" / "bad.json").write_text("{")
    assert F.load_traffic("bad.json", "msm")[0] is None and F.load_profiled("bad.json", "msm", "x") is None


def test_the_real_source_lists_exist():
    """a renamed kernel source must not silently drop out of the fingerprint (the .cuh -> .hip.h rename of round 4)"""
    from tools.fingerprint import SOURCES
    for which, files in SOURCES.items():
        for rel in files:
            assert os.path.exists(os.path.join(ROOT, rel)), (which, rel)


def test_committed_bench_line_keeps_the_contract():
    """the default bench line of the round's final build (profiles/r04_bench_default.json, written by bench.py on the GPU box): the
    keys the driver and the judge read are there, the roofline arithmetic is consistent, the profiled kernel average agrees with the
    run's own event timing to a few per cent (round 3: 13-15 % apart), and the proofs in `extra` matched their committed closed forms"""
    path = os.path.join(ROOT, "profiles", "r04_bench_default.json")
    with open(path) as f:
        line = json.loads(f.read().strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["unit"] == "Mscalar/s" and line["vs_baseline"] is None and "workload" in line["config"]
    assert abs(line["value"] - (1 << 20) / line["ms_per_step"] / 1e3) / line["value"] < 0.01
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6
    assert abs(r["achieved"] - 96 * (1 << 20) / (r["kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 0.01     # 96 B per pair, SURVEY 8(d)
    assert r["traffic"] and abs(r["frac_measured"] - r["traffic"] / (r["kernel_ms"] * 1e-3) / 1e9 / 8000.0) < 1e-3
    assert abs(r["kernel_ms_profiled"] - r["kernel_ms"]) / r["kernel_ms"] < 0.05
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "Mscalar/s" and c["sample"]
    ex = line["extra"]
    for key in ("groth16_prove_bn254_2^20", "groth16_prove_bls12_381_2^23"):
        assert ex[key]["verifies"] and ex[key]["matches_committed_closed_form"]
    assert ex["groth16_prove_bn254_2^20"]["cpu_baseline"]["kind"] == "port"
