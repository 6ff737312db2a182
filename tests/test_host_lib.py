"""CPU: the C-ABI shared library loads, exports every symbol include/zkmi.h declares, and its host-side
functions (single-point arithmetic, codecs, scalar helpers) agree with the oracle.  No GPU compute."""

import os
import random
import re

import numpy as np
import pytest

from oracle import corc, pyref as R
from zksnake_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CURVES = (("BN254", 0), ("BLS12_381", 1))


def test_header_symbols_are_exported_and_bound(lib):
    with open(os.path.join(ROOT, "include", "zkmi.h")) as f:
        text = f.read()
    declared = set(re.findall(r"\b(zk_[a-z0-9_]+)\s*\(", text))
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/zkmi.h but not exported by libzkmi.so"
    assert declared == set(N.SIGNATURES), declared ^ set(N.SIGNATURES)
    assert lib.zk_version().startswith(b"zkmi")


def test_sizes(lib):
    assert [lib.zk_fq_limbs(c) for c in (0, 1, 2)] == [4, 6, -1]
    assert [lib.zk_point_limbs(0, 1), lib.zk_point_limbs(0, 2), lib.zk_point_limbs(1, 1), lib.zk_point_limbs(1, 2)] == [8, 16, 12, 24]
    assert [lib.zk_point_bytes(0, 1), lib.zk_point_bytes(0, 2), lib.zk_point_bytes(1, 1), lib.zk_point_bytes(1, 2)] == [32, 64, 48, 96]


def test_no_gpu_fails_loudly(lib):
    """without a GPU the compute entry points must report ZK_ERR_HIP, never fall back to the CPU"""
    if lib.zk_device_count() > 0:
        pytest.skip("a GPU is visible")
    assert lib.zk_init(0) == N.ZK_ERR_HIP
    assert b"no CPU fallback" in lib.zk_last_error()
    a = np.zeros((4, 4), dtype=np.uint64)
    out = np.zeros((4, 4), dtype=np.uint64)
    assert lib.zk_ntt(0, 0, 0, 4, N.u64p(a), 4, N.u64p(out)) == N.ZK_ERR_HIP
    bases = np.zeros((4, 8), dtype=np.uint64)
    res = np.zeros(8, dtype=np.uint64)
    assert lib.zk_msm(0, 1, 4, 4, N.u64p(a), N.u64p(bases), N.u64p(res)) == N.ZK_ERR_HIP
    with pytest.raises(N.ZkError):
        N.ensure_gpu()


def _queue_probe(code, env_extra):
    """run `code` in a fresh interpreter (the decision is taken once per process) and return its stdout lines"""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "ZKMI_TEST_RUNTIME_STARTED", "ZKMI_HW_QUEUES_SET_BY_LIBRARY")}
    env.update(env_extra)
    env["PYTHONPATH"] = ROOT
    res = subprocess.run([sys.executable, "-W", "always", "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    return res.stdout.split(), res.stderr


_QUEUE_CODE = """
import ctypes
{first}
from zksnake_amd import _native as N
lib = N.load()
{second}
q = ctypes.c_int(-1)
st = lib.zk_hw_queues_prepare(ctypes.byref(q))
libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p
st2, q2 = ctypes.c_int(-1), ctypes.c_int(-1)
rc = lib.zk_init_ex(0, ctypes.byref(st2), ctypes.byref(q2))
print(st, q.value, (libc.getenv(b"GPU_MAX_HW_QUEUES") or b"-").decode(), N.queue_status, st2.value, q2.value)
"""


@pytest.mark.parametrize("order", ["torch_first", "lib_first"])
def test_hw_queues_are_set_by_the_library_in_either_import_order(order):
    """GPU_MAX_HW_QUEUES is put in place by the C library (zk_hw_queues_prepare / zk_init), not by a Python import side
    effect: unset -> the library sets 12 before its first HIP call, whichever of torch and the library is imported first
    (importing torch does not start the HIP runtime); the status is the same from every entry point"""
    first, second = ("import torch", "") if order == "torch_first" else ("", "import torch")
    out, _ = _queue_probe(_QUEUE_CODE.format(first=first, second=second), {})
    assert out == [str(N.QUEUES_SET_BY_LIBRARY), "12", "12", str(N.QUEUES_SET_BY_LIBRARY), str(N.QUEUES_SET_BY_LIBRARY), "12"]


def test_hw_queues_caller_setting_wins_and_late_start_is_reported():
    out, _ = _queue_probe(_QUEUE_CODE.format(first="", second=""), {"GPU_MAX_HW_QUEUES": "4"})
    assert out == [str(N.QUEUES_CALLER), "4", "4", str(N.QUEUES_CALLER), str(N.QUEUES_CALLER), "4"]
    # the runtime already running (a C-ABI consumer whose host initialised HIP first): reported, warned about, nothing set
    out, err = _queue_probe(_QUEUE_CODE.format(first="", second=""), {"ZKMI_TEST_RUNTIME_STARTED": "1"})
    assert out == [str(N.QUEUES_TOO_LATE), "0", "-", str(N.QUEUES_TOO_LATE), str(N.QUEUES_TOO_LATE), "0"]
    assert "GPU_MAX_HW_QUEUES" in err and "RuntimeWarning" in err


@pytest.mark.parametrize("name,cid", CURVES)
@pytest.mark.parametrize("grp", [1, 2])
def test_point_arithmetic_and_codec(lib, name, cid, grp):
    cv = R.curve_by_name(name)
    g = R.Group(cv, grp)
    rnd = random.Random(5)
    W = N.point_limbs(cid, grp)
    gen = np.zeros(W, dtype=np.uint64)
    N.check(lib.zk_point_generator(cid, grp, N.u64p(gen)))
    assert corc.limbs_to_points(gen, cid, grp)[0] == g.gen and lib.zk_point_on_curve(cid, grp, N.u64p(gen)) == 1
    pts = []
    for k in [0, 1, 2, cv.r - 1, cv.r, cv.r + 5] + [rnd.randrange(cv.r) for _ in range(4)]:
        out = np.zeros(W, dtype=np.uint64)
        N.check(lib.zk_point_mul(cid, grp, N.u64p(gen), N.u64p(N.ints_to_limbs([k])), N.u64p(out)))
        P = corc.limbs_to_points(out, cid, grp)[0]
        assert P == g.mul(g.gen, k)
        pts.append((out, P))
    nb = lib.zk_point_bytes(cid, grp)
    for a, Pa in pts:
        for b, Pb in pts[:5]:
            out = np.zeros(W, dtype=np.uint64)
            N.check(lib.zk_point_add(cid, grp, N.u64p(a), N.u64p(b), N.u64p(out)))
            assert corc.limbs_to_points(out, cid, grp)[0] == g.add(Pa, Pb)
        buf = np.zeros(nb, dtype=np.uint8)
        N.check(lib.zk_point_compress(cid, grp, N.u64p(a), N.u8p(buf)))
        assert bytes(buf) == R.compress(cv, grp, Pa)
        back = np.zeros(W, dtype=np.uint64)
        N.check(lib.zk_point_decompress(cid, grp, N.u8p(buf), N.u64p(back)))
        assert (back == a).all()
        neg = np.zeros(W, dtype=np.uint64)
        N.check(lib.zk_point_neg(cid, grp, N.u64p(a), N.u64p(neg)))
        assert corc.limbs_to_points(neg, cid, grp)[0] == g.neg(Pa)
    # rejected encodings: bad flags, x off the curve
    back = np.zeros(W, dtype=np.uint64)
    bad = np.frombuffer(bytes([0xFF] * nb), dtype=np.uint8).copy()
    assert lib.zk_point_decompress(cid, grp, N.u8p(bad), N.u64p(back)) == N.ZK_ERR_POINT
    off = np.zeros(W, dtype=np.uint64)
    off[0] = 5
    assert lib.zk_point_on_curve(cid, grp, N.u64p(off)) == 0


def test_golden_encodings(lib):
    import json
    with open(os.path.join(ROOT, "tests", "golden", "oracle_vectors.json")) as f:
        gold = json.load(f)
    for name, cid in CURVES:
        for grp in (1, 2):
            og = gold[name][f"g{grp}"]
            W = N.point_limbs(cid, grp)
            nb = lib.zk_point_bytes(cid, grp)
            gen = np.zeros(W, dtype=np.uint64)
            N.check(lib.zk_point_generator(cid, grp, N.u64p(gen)))
            buf = np.zeros(nb, dtype=np.uint8)
            N.check(lib.zk_point_compress(cid, grp, N.u64p(gen), N.u8p(buf)))
            assert bytes(buf).hex() == og["generator_compressed"]
            N.check(lib.zk_point_compress(cid, grp, N.u64p(np.zeros(W, dtype=np.uint64)), N.u8p(buf)))
            assert bytes(buf).hex() == og["infinity_compressed"]
            # dlog * G compresses to the recorded MSM result
            out = np.zeros(W, dtype=np.uint64)
            N.check(lib.zk_point_mul(cid, grp, N.u64p(gen), N.u64p(N.ints_to_limbs([int(og["msm_dlog"])])), N.u64p(out)))
            N.check(lib.zk_point_compress(cid, grp, N.u64p(out), N.u8p(buf)))
            assert bytes(buf).hex() == og["msm_compressed"]


@pytest.mark.parametrize("name,cid", CURVES)
def test_scalar_field_helpers(lib, name, cid):
    cv = R.curve_by_name(name)
    rnd = random.Random(6)
    for n in (1, 2, 8, 64, 1 << 20):
        out = np.zeros(4, dtype=np.uint64)
        N.check(lib.zk_fr_root_of_unity(cid, n, N.u64p(out)))
        assert N.limbs_to_ints(out.reshape(1, 4))[0] == cv.root_of_unity(n)
    for n in (1, 2, 8, 64):
        for tau in (rnd.randrange(cv.r), cv.root_of_unity(n) if n > 1 else 1, 1, cv.r + 3):
            o = np.zeros((n, 4), dtype=np.uint64)
            N.check(lib.zk_fr_lagrange_coeffs(cid, n, N.u64p(N.ints_to_limbs([tau])), N.u64p(o)))
            assert N.limbs_to_ints(o) == R.lagrange_at(n, tau % cv.r, cv)
    out = np.zeros(4, dtype=np.uint64)
    assert lib.zk_fr_root_of_unity(cid, 1 << 40, N.u64p(out)) == N.ZK_ERR_DOMAIN


def test_relaxed_g2_step_host_build_and_integer_models(tmp_path):
    """the relaxed-range G2 mixed addition of the accumulate kernel: (1) a host build of the library's own field / curve code
    runs it against the plain formulas on chains of additions with the special cases (tests/native/relaxed_g2_check.cpp);
    (2) the integer models replay its limb operations, and those of the lazy NTT butterfly, with every limb, column-sum
    and value bound asserted on inputs pushed to the ends of their ranges (tools/model_relaxed_g2.py, tools/model_lazy_ntt.py)"""
    import subprocess
    import sys
    exe = str(tmp_path / "relaxed_g2_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "relaxed_g2_check.cpp")])
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and res.stdout.count(": ok") == 2, res.stdout + res.stderr
    for model in ("model_relaxed_g2.py", "model_lazy_ntt.py"):
        res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", model)], capture_output=True, text=True, timeout=600)
        assert res.returncode == 0 and "ok" in res.stdout, res.stdout + res.stderr


def test_build_script_falls_back_to_plain_hipcc_when_the_pass_pipeline_cannot_be_reproduced(tmp_path):
    """csrc/hipcc_noreassoc.sh drives clang / opt / llc by hand to leave LLVM's `reassociate` pass out of the device pipeline; on a
    toolchain whose O3 pipeline text has no such pass it must not fail the build (round-3 advisor finding) but compile the unit
    with the stock driver, LLCFLAGS travelling as -mllvm options.  Fake tools stand in for the toolchain."""
    import stat
    import subprocess
    script = os.path.join(ROOT, "zksnake_amd", "csrc", "hipcc_noreassoc.sh")
    llvm = tmp_path / "llvm"
    llvm.mkdir()

    def tool(path, body):
        path.write_text("#!/bin/bash\n" + body)
        path.chmod(path.stat().st_mode | stat.S_IXUSR)

    tool(llvm / "clang++", 'while [ $# -gt 0 ]; do if [ "$1" = "-o" ]; then : > "$2"; fi; shift; done\n')
    tool(llvm / "opt", 'echo "module(function(instcombine,simplifycfg)),BitcodeWriterPass"; exit 1\n')   # no reassociate in this pipeline
    hipcc = tmp_path / "hipcc"
    tool(hipcc, f'echo "$@" > {tmp_path}/hipcc_args; while [ $# -gt 0 ]; do if [ "$1" = "-o" ]; then echo obj > "$2"; fi; shift; done\n')
    src = tmp_path / "unit.hip"
    src.write_text("// nothing\n")
    out = tmp_path / "unit.o"
    env = dict(os.environ, LLVM_BIN=str(llvm), HIPCC=str(hipcc), ARCH="gfx950", LLCFLAGS="-amdgpu-sched-strategy=max-ilp", TMPDIR=str(tmp_path))
    res = subprocess.run(["bash", script, str(out), str(src), "-O3", "-DZK_GROUP=Bn254G1"], env=env, capture_output=True, text=True, timeout=60)
    assert res.returncode == 0, res.stderr
    assert "plain hipcc" in res.stderr
    args = (tmp_path / "hipcc_args").read_text()
    assert "--offload-arch=gfx950" in args and "-mllvm -amdgpu-sched-strategy=max-ilp" in args and "-DZK_GROUP=Bn254G1" in args
    assert out.read_text() == "obj\n"
