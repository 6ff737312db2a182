#!/usr/bin/env python3
"""Turn rocprofv3 output under gpurun_out/ into the committed summaries under profiles/.

  python tools/summarize_profiles.py <round-tag> <kernel-trace-dir> [<fetch-pmc-dir> <write-pmc-dir>]

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats summary, verbatim),
profiles/<tag>_pmc_summary.json (FETCH_SIZE / WRITE_SIZE per kernel, raw and corrected) and
profiles/msm_traffic.json (the accumulate kernel's HBM bytes per launch that bench.py reports as
roofline.traffic).  Correction per MI355X_MICROARCH.md "HBM": on gfx950 FETCH_SIZE counts half of the bytes
of 16-byte-per-lane loads (verified here on bases_to_mont_kernel, which reads exactly n*64 bytes);
WRITE_SIZE is exact for 16-byte stores.  Both counters are in KiB.
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


def main():
    tag, kt = sys.argv[1], sys.argv[2]
    prof = os.path.join(ROOT, "profiles")
    stats = glob.glob(os.path.join(kt, "**", "*_kernel_stats.csv"), recursive=True)[0]
    shutil.copy(stats, os.path.join(prof, f"{tag}_kernel_stats.csv"))
    if len(sys.argv) >= 5:
        fetch_csv = glob.glob(os.path.join(sys.argv[3], "**", "*_counter_collection.csv"), recursive=True)[0]
        write_csv = glob.glob(os.path.join(sys.argv[4], "**", "*_counter_collection.csv"), recursive=True)[0]
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
        with open(os.path.join(prof, f"{tag}_pmc_summary.json"), "w") as f:
            json.dump(summary, f, indent=1, sort_keys=True)
        key = next((k for k in summary if k.startswith("zkmi::accumulate_kernel<zkmi::Bn254G1>")), None)
        if key:
            with open(os.path.join(prof, "msm_traffic.json"), "w") as f:
                json.dump({
                    "accumulate_hbm_bytes_per_launch": summary[key]["hbm_bytes_per_launch_corrected"],
                    "source": f"profiles/{tag}_pmc_summary.json: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on "
                              "`bench.py --steps 3`, (2*FETCH_SIZE + WRITE_SIZE) KiB per launch (gfx950 half-count correction)",
                }, f, indent=1)
    print("profiles updated for", tag)


if __name__ == "__main__":
    main()
