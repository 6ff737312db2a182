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


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                acc[row["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def first(pattern):
    hits = glob.glob(pattern, recursive=True)
    return hits[0] if hits else None


def main():
    tag, prefix = sys.argv[1], sys.argv[2]
    prof = os.path.join(ROOT, "profiles")
    out_root = os.path.join(ROOT, "gpurun_out")
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
                with open(os.path.join(prof, "msm_traffic.json"), "w") as f:
                    json.dump({
                        "accumulate_hbm_bytes_per_launch": summary[key]["hbm_bytes_per_launch_corrected"],
                        "source": f"profiles/{tag}_bench_pmc_summary.json: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on "
                                  "`bench.py --steps 3`, (2*FETCH_SIZE + WRITE_SIZE) KiB per launch (gfx950 half-count correction)",
                    }, f, indent=1)
        print(w, "summarised")
    print("profiles updated for", tag)


if __name__ == "__main__":
    main()
