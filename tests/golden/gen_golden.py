#!/usr/bin/env python3
"""
Generates tests/golden/*.json.  Run in the build container (needs /root/reference for part 1):

    python tests/golden/gen_golden.py

Part 1 -- outputs of the REFERENCE ITSELF: the three pure-Python modules of the reference that import
without its Rust extension (zksnake/utils.py, zksnake/array.py, zksnake/constant.py) are
loaded from /root/reference by file path and run on seeded inputs.  Only inputs and outputs are stored.

Part 2 -- vectors from the definitional oracle (oracle/pyref.py) for the parts of the path whose reference
implementation (arkworks, Rust) cannot run here: NTT vectors, MSM results, compressed encodings and
Groth16 proofs with pinned toxic waste.  These pin the oracle and the product against silent drift;
they are NOT outputs of the reference binary (parity unpinned).
"""
import importlib.util
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/python/zksnake"


def load_ref(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def part1():
    utils, array, const = load_ref("utils"), load_ref("array"), load_ref("constant")
    rnd = random.Random(0xA11CE)
    out = {"constants": {k: str(getattr(const, k)) for k in
                         ("BN254_MODULUS", "BN254_SCALAR_FIELD", "BLS12_381_MODULUS", "BLS12_381_SCALAR_FIELD")}}
    out["next_power_of_two"] = [[n, utils.next_power_of_two(n)] for n in (1, 2, 3, 4, 5, 7, 8, 9, 1000, 1 << 20, (1 << 20) + 1)]
    out["is_power_of_two"] = [[n, utils.is_power_of_two(n)] for n in (1, 2, 3, 4, 6, 8, 1023, 1024)]
    p = const.BN254_SCALAR_FIELD
    a = [rnd.randrange(1, p) for _ in range(9)]
    out["batch_modinv"] = {"m": str(p), "a": [str(x) for x in a], "out": [str(x) for x in utils.batch_modinv(a, p)]}
    b = [rnd.randrange(p) for _ in range(9)]
    out["inner_product"] = {"a": [str(x) for x in a], "b": [str(x) for x in b], "p": str(p), "out": str(utils.inner_product(a, b, p))}
    out["split_list"] = {"data": list(range(10)), "n": 4, "out": utils.split_list(list(range(10)), 4)}
    # SparseArray: dense constructor, append, dot
    dense = [[rnd.randrange(3) * rnd.randrange(p) for _ in range(5)] for _ in range(4)]
    sa = array.SparseArray(dense, 4, 5, p)
    extra = [(rnd.randrange(4), rnd.randrange(5), rnd.randrange(p)) for _ in range(6)] + [(0, 0, 0)]
    sa.append(extra)
    vec = [rnd.randrange(p) for _ in range(5)]
    out["sparse_array"] = {
        "p": str(p), "dense": [[str(v) for v in row] for row in dense], "append": [[r, c, str(v)] for r, c, v in extra],
        "vector": [str(v) for v in vec], "triplets": [[r, c, str(v)] for r, c, v in sa.triplets],
        "triplets_map": {str(k): [[c, str(v)] for c, v in lst] for k, lst in sa.triplets_map.items()},
        "dot": [str(v) for v in sa.dot(vec)],
    }
    return out


def part2():
    from oracle import pyref as R
    rnd = random.Random(0xB0B)
    out = {}
    for name in ("BN254", "BLS12_381"):
        cv = R.curve_by_name(name)
        o = {}
        vals = [rnd.randrange(cv.r) for _ in range(16)]
        o["ntt16"] = {"in": [str(v) for v in vals], "fwd": [str(v) for v in R.ntt(vals, 16, cv)],
                      "inv": [str(v) for v in R.ntt(vals, 16, cv, inverse=True)],
                      "coset_fwd": [str(v) for v in R.coset_ntt(vals, 16, cv)]}
        o["root_of_unity"] = {str(n): str(cv.root_of_unity(n)) for n in (2, 4, 1 << 10, 1 << 20, 1 << 22)}
        for grp in (1, 2):
            g = R.Group(cv, grp)
            ks = [rnd.randrange(cv.r) for _ in range(6)]
            sc = [rnd.randrange(cv.r) for _ in range(6)]
            sc[0], sc[1] = 0, cv.r - 1
            pts = [g.mul(g.gen, k) for k in ks]
            res = g.msm(pts, sc)
            o[f"g{grp}"] = {
                "generator_compressed": R.compress(cv, grp, g.gen).hex(),
                "infinity_compressed": R.compress(cv, grp, None).hex(),
                "base_logs": [str(k) for k in ks], "scalars": [str(s) for s in sc],
                "bases_compressed": [R.compress(cv, grp, P).hex() for P in pts],
                "msm_compressed": R.compress(cv, grp, res).hex(),
                "msm_dlog": str(sum(a * b for a, b in zip(ks, sc)) % cv.r),
            }
        toxic, rs = (11, 22, 33, 44, 55), (66, 77)
        proofs = {}
        for cname, circ in (("readme", R.readme_circuit(cv.r)), ("chain8", R.chain_circuit(8, cv.r))):
            A, B, C, n_row, n_col, n_pub, w = circ
            a, b, c = R.groth16_closed_form(A, B, C, n_row, n_col, n_pub, w, cv, toxic, rs)
            g1, g2 = R.G1(cv), R.G2(cv)
            pr = (g1.mul(g1.gen, a), g2.mul(g2.gen, b), g1.mul(g1.gen, c))
            proofs[cname] = {"a": str(a), "b": str(b), "c": str(c), "proof_bytes": R.proof_bytes(cv, pr).hex()}
        o["groth16"] = {"toxic": list(toxic), "blinding": list(rs), "proofs": proofs}
        out[name] = o
    return out


if __name__ == "__main__":
    with open(os.path.join(HERE, "reference_pure_python.json"), "w") as f:
        json.dump(part1(), f, indent=1, sort_keys=True)
    with open(os.path.join(HERE, "oracle_vectors.json"), "w") as f:
        json.dump(part2(), f, indent=1, sort_keys=True)
    print("wrote golden fixtures")
