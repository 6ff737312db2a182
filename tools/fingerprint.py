"""Fingerprint of the kernel sources a PMC-derived traffic figure belongs to.

`profiles/msm_traffic.json` / `profiles/prove_traffic.json` are written on the GPU box by tools/summarize_profiles.py
together with this fingerprint; bench.py reports their figures as `roofline.traffic` only while the fingerprint of the
sources it runs from is the same, so a figure cannot silently outlive the kernel it was measured on."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SOURCES = {
    # the accumulate kernel and everything inlined into it
    "msm": ("zksnake_amd/csrc/msm_impl.hip.h", "zksnake_amd/csrc/msm_accumulate.hip.h", "zksnake_amd/csrc/msm_common.hip.h",
            "zksnake_amd/csrc/msm_sort.hip.h", "zksnake_amd/csrc/msm_reduce.hip.h", "zksnake_amd/csrc/curve.hip.h",
            "zksnake_amd/csrc/field.hip.h", "zksnake_amd/csrc/field_params.h", "zksnake_amd/csrc/msm_plan.h",
            "zksnake_amd/csrc/pair.hip.h", "zksnake_amd/csrc/fp2_split.hip.h", "zksnake_amd/csrc/hipcc_noreassoc.sh", "zksnake_amd/csrc/Makefile"),
}
SOURCES["prove"] = SOURCES["msm"] + ("zksnake_amd/csrc/ntt.hip", "zksnake_amd/csrc/fr_mem.hip.h", "zksnake_amd/groth16/protocol.py",
                                      "zksnake_amd/groth16/qap.py")


def source_fingerprint(which="msm"):
    h = hashlib.sha256()
    for rel in SOURCES[which]:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(rel.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()[:16]


def load_profiled(name, which, key):
    """another field of profiles/<name> (e.g. "kernel_avg_ms_profiled") under the same fingerprint rule, or None"""
    import json
    path = os.path.join(ROOT, "profiles", name)
    try:
        with open(path) as f:
            tj = json.load(f)
    except Exception:  # noqa: BLE001 - absent or unreadable: nothing to report
        return None
    return tj.get(key) if tj.get("source_fingerprint") == source_fingerprint(which) else None


def load_traffic(name, which):
    """(bytes or None, source string): the figure of profiles/<name> if it was measured on the present sources"""
    import json
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, None
    try:
        with open(path) as f:
            tj = json.load(f)
    except Exception:  # noqa: BLE001
        return None, "unreadable " + name
    want = source_fingerprint(which)
    if tj.get("source_fingerprint") != want:
        return None, (f"profiles/{name} was measured on other kernel sources (fingerprint {tj.get('source_fingerprint')}, now {want}): "
                      "not reported; re-run tools/collect_profiles.sh + tools/summarize_profiles.py")
    return tj.get("hbm_bytes"), tj.get("source")
