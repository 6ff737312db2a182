#!/usr/bin/env python3
"""Turn the rocprofv3 output of tools/collect_profiles.sh (under gpurun_out/) into the committed summaries under profiles/.

  python tools/summarize_profiles.py <round-tag> <prefix>        e.g.  r02a p4

For every workload <w> in (bench, groups, ntt, prove, plonk) with gpurun_out/<prefix>_<w>_kt present:
  profiles/<tag>_<w>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, verbatim
  profiles/<tag>_<w>_pmc_summary.json   FETCH_SIZE / WRITE_SIZE per kernel (separate --pmc passes), raw and corrected
and profiles/msm_traffic.json (the accumulate kernel's HBM bytes per launch that bench.py reports as roofline.traffic).
Correction per MI355X_MICROARCH.md "HBM": on gfx950 FETCH_SIZE counts half of the bytes of 16-byte-per-lane loads
(verified on bases_to_mont_kernel, which reads exactly n * 64 bytes); WRITE_SIZE is exact for 16-byte stores.  Both counters
are in KiB.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.fingerprint import source_fingerprint  # noqa: E402


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                acc[row["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def first(pattern):
    hits = glob.glob(pattern, recursive=True)
    # gpurun merges every collection into the same directories (the files carry the profiler's process id): take the newest
    return max(hits, key=os.path.getmtime) if hits else None


def kernel_average_ns(stats_csv, name_prefix):
    """AverageNs (and Calls) of the first kernel of a rocprofv3 *_kernel_stats.csv whose name starts with `name_prefix`"""
    with open(stats_csv) as f:
        for row in csv.DictReader(f):
            if row["Name"].replace("void ", "").startswith(name_prefix):
                return float(row["AverageNs"]), int(row["Calls"])
    return None, 0


def main(tag=None, prefix=None, out_root=None, prof=None):
    tag, prefix = tag or sys.argv[1], prefix or sys.argv[2]
    prof = prof or os.path.join(ROOT, "profiles")
    out_root = out_root or os.path.join(ROOT, "gpurun_out")
    for w in ("bench", "groups", "ntt", "prove", "plonk"):
        stats = first(os.path.join(out_root, f"{prefix}_{w}_kt", "**", "*_kernel_stats.csv"))
        if not stats:
            continue
        shutil.copy(stats, os.path.join(prof, f"{tag}_{w}_kernel_stats.csv"))
        fetch_csv = first(os.path.join(out_root, f"{prefix}_{w}_fetch", "**", "*_counter_collection.csv"))
        write_csv = first(os.path.join(out_root, f"{prefix}_{w}_write", "**", "*_counter_collection.csv"))
        if not (fetch_csv and write_csv):
            continue
        fetch, calls = per_kernel(fetch_csv, "FETCH_SIZE")
        write, _ = per_kernel(write_csv, "WRITE_SIZE")
        summary = {}
        for k in sorted(set(fetch) | set(write)):
            f_kib, w_kib = fetch.get(k, 0.0), write.get(k, 0.0)
            summary[k] = {
                "launches_sampled": calls.get(k, 0),
                "FETCH_SIZE_KiB_raw": round(f_kib, 1),
                "WRITE_SIZE_KiB": round(w_kib, 1),
                "hbm_bytes_per_launch_corrected": int((2 * f_kib + w_kib) * 1024),
            }
        with open(os.path.join(prof, f"{tag}_{w}_pmc_summary.json"), "w") as f:
            json.dump(summary, f, indent=1, sort_keys=True)
        if w == "bench":
            key = next((k for k in summary if k.startswith("zkmi::accumulate_kernel<zkmi::Bn254G1>")), None)
            if key:
                # the dominant kernel's average duration in the kernel-trace pass of the SAME command: bench.py prints it as
                # kernel_ms_profiled beside its own event timing, so a profile that does not reproduce the line shows in the record
                avg_ns, calls = kernel_average_ns(stats, key)
                with open(os.path.join(prof, "msm_traffic.json"), "w") as f:
                    json.dump({
                        "hbm_bytes": summary[key]["hbm_bytes_per_launch_corrected"],
                        "kernel": key,
                        "kernel_avg_ms_profiled": round(avg_ns / 1e6, 4) if avg_ns else None,
                        "kernel_launches_profiled": calls,
                        "source_fingerprint": source_fingerprint("msm"),
                        "source": f"profiles/{tag}_bench_pmc_summary.json: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on "
                                  "`bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra` (the driver's command), (2*FETCH_SIZE + "
                                  f"WRITE_SIZE) KiB per launch (gfx950 half-count correction); kernel average from profiles/{tag}_bench_kernel_stats.csv",
                    }, f, indent=1)
        if w == "prove":
            # one steady-state proof = every kernel of the prove path once per proof; setup-only kernels are left out.
            # proofs in the run = launches of the G2 accumulate kernel (one per proof)
            proofs = next((v["launches_sampled"] for k, v in summary.items() if k.startswith("zkmi::accumulate_kernel<zkmi::Bn254G2>")), 0)
            setup_only = ("fixed_mul_kernel", "fixed_table_kernel", "normalize_kernel", "dbl_rows_kernel", "varbase_mul_kernel", "powers_kernel",
                          "twiddle_kernel", "bases_to_mont_kernel", "vec_axpby", "points_encode", "points_decode")
            if proofs:
                total = sum(v["hbm_bytes_per_launch_corrected"] * v["launches_sampled"] for k, v in summary.items()
                            if not any(s in k for s in setup_only))
                with open(os.path.join(prof, "prove_traffic.json"), "w") as f:
                    json.dump({
                        "hbm_bytes": int(total / proofs),
                        "proofs_in_run": proofs,
                        "source_fingerprint": source_fingerprint("prove"),
                        "source": f"profiles/{tag}_prove_pmc_summary.json: FETCH_SIZE / WRITE_SIZE passes over tools/prove_bench.py --log-n 20, sum over the "
                                  "prove-path kernels of (2*FETCH_SIZE + WRITE_SIZE) KiB x launches, divided by the proofs in the run (setup-only "
                                  "kernels left out; the matrices' SpMV of the first proof's upload is included once per proof)",
                    }, f, indent=1)
        print(w, "summarised")
    lds_csv = first(os.path.join(out_root, f"{prefix}_ntt_lds", "**", "*_counter_collection.csv"))
    if lds_csv:
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        with open(lds_csv) as f:
            for row in csv.DictReader(f):
                acc[row["Kernel_Name"].split("(")[0].replace("void ", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
        lds = {}
        for k, ctrs in sorted(acc.items()):
            if "ntt_pass_kernel" not in k:
                continue
            # the launches of one transform differ (first / middle / last pass): keep them apart by position in the cycle of three
            n_l = max(len(v) for v in ctrs.values())
            passes = []
            for ph in range(3):
                d = {c: sum(v[ph::3]) / max(1, len(v[ph::3])) for c, v in ctrs.items()}
                if d.get("SQ_LDS_IDX_ACTIVE") and d.get("SQ_WAVE_CYCLES"):
                    d = {c: round(x, 1) for c, x in d.items()}
                    d["bank_conflict_share_of_lds_active_cycles"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"], 4)
                    d["wave_cycles_waiting_on_lds_share"] = round(d.get("SQ_WAIT_INST_LDS", 0.0) / d["SQ_WAVE_CYCLES"], 4)
                passes.append(d)
            lds[k] = {"launches": n_l, "pass_0_1_2_of_a_2^22_transform": passes}
        with open(os.path.join(prof, f"{tag}_ntt_pmc_lds.json"), "w") as f:
            json.dump({"command": "rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES "
                                  "SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT --kernel-trace (tools/collect_profiles.sh runlds)", "kernels": lds}, f, indent=1)
        print("ntt lds counters summarised")
    for w in ("bench", "ntt"):
        sq_csv = first(os.path.join(out_root, f"{prefix}_{w}_sq", "**", "*_counter_collection.csv"))
        if not sq_csv:
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        with open(sq_csv) as f:
            for row in csv.DictReader(f):
                acc[row["Kernel_Name"].split("(")[0].replace("void ", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
        sq = {}
        for k, ctrs in sorted(acc.items()):
            sq[k] = {c: round(sum(v) / len(v), 1) for c, v in sorted(ctrs.items())}
            sq[k]["launches"] = max(len(v) for v in ctrs.values())
            wc, act, wi = sq[k].get("SQ_WAVE_CYCLES"), sq[k].get("SQ_ACTIVE_INST_VALU"), sq[k].get("SQ_WAIT_INST_ANY")
            if wc and sq[k].get("SQ_WAVES"):
                sq[k]["valu_instructions_per_wave"] = round(sq[k].get("SQ_INSTS_VALU", 0.0) / sq[k]["SQ_WAVES"], 1)
                # SQ_WAVE_CYCLES, SQ_WAIT_* and SQ_ACTIVE_* count in units of 4 cycles per wave: shares of a wave's residency
                sq[k]["share_of_wave_cycles"] = {"valu_executing": round((act or 0.0) / wc, 3), "waiting_to_issue": round((wi or 0.0) / wc, 3)}
        with open(os.path.join(prof, f"{tag}_{w}_pmc_sq.json"), "w") as f:
            json.dump({"command": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU "
                                  "SQ_BUSY_CYCLES SQ_WAVES --kernel-trace (tools/collect_profiles.sh runsq)", "kernels": sq}, f, indent=1)
        print(w, "sq counters summarised")
    print("profiles updated for", tag)


if __name__ == "__main__":
    main()
