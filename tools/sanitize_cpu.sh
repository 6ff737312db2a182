#!/bin/bash
# CPU-side sanitizer pass (AddressSanitizer + UndefinedBehaviorSanitizer) over everything that runs on the host:
#   * the host half of libzkmi.so (point operations, pairing, codecs, MSM tail, plan bookkeeping) compiled HOST-ONLY from the
#     same sources (--offload-host-only: no device code, so nothing here can launch a kernel),
#   * the CPython marshalling helper (_pyints.so, thread pool included),
#   * the oracle (test infrastructure),
# then the CPU test-suite files that exercise them, run against those builds.  GPU sanitizers are not available on the pool;
# this is the part of the product a sanitizer can see.  Usage: bash tools/sanitize_cpu.sh [pytest args]
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/build/san"
CSRC="$ROOT/zksnake_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
CLANG_RT="$(dirname "$($HIPCC --print-file-name=libclang_rt.asan-x86_64.so 2>/dev/null || true)")"
[ -f "$CLANG_RT/libclang_rt.asan-x86_64.so" ] || CLANG_RT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g"
mkdir -p "$OUT/obj" "$OUT/pkg"

HOSTFLAGS="-O1 -std=c++17 --offload-host-only --offload-arch=gfx950 -fPIC -Wno-unused-function -Wno-pass-failed -mbmi2 -madx -DZK_NOINLINE_MUL $SAN"
pids=()
build() { # build <object> <source> [defines]
    local obj="$OUT/obj/$1.o"; shift
    local src="$1"; shift
    if [ ! -f "$obj" ] || [ -n "$(find "$CSRC" "$ROOT/include" -newer "$obj" -type f | head -1)" ]; then
        $HIPCC $HOSTFLAGS "$@" -c "$src" -o "$obj" &
        pids+=($!)
        # eight compilers at a time (the container has eight cores)
        if [ ${#pids[@]} -ge 8 ]; then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
    fi
}
for f in host pairing ntt plonk msm; do build "$f" "$CSRC/$f.hip"; done
for g in Bn254G1 Bn254G2 Bls381G1 Bls381G2; do
    for p in 0 1 2 3; do build "msm_group_${g}_$p" "$CSRC/msm_group.hip" -DZK_GROUP=$g -DZK_PART=$p; done
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
# a host-only object still refers to its translation unit's device image (__hip_fatbin_<hash>): give each an EMPTY offload bundle
# (the magic string and a bundle count of zero), which the HIP runtime registers at load time and never has to look into
rm -f "$OUT/obj/fatbin_stub.o"
{
    echo '#include <stdint.h>'
    echo 'struct bundle { char magic[24]; uint64_t count; };'
    nm -u "$OUT"/obj/*.o | grep -o '__hip_fatbin_[0-9a-f]*' | sort -u | while read -r sym; do
        echo "extern \"C\" __attribute__((aligned(4096))) const bundle $sym = {{'_','_','C','L','A','N','G','_','O','F','F','L','O','A','D','_','B','U','N','D','L','E','_','_'}, 0};"
    done
} > "$OUT/fatbin_stub.cpp"
g++ -O0 -fPIC -c "$OUT/fatbin_stub.cpp" -o "$OUT/obj/fatbin_stub.o"
$HIPCC --offload-host-only -shared -fPIC $SAN -shared-libsan -o "$OUT/libzkmi.so" "$OUT"/obj/*.o

PYINC=$(python3 -c "import sysconfig; print(sysconfig.get_paths()['include'])")
# the helper is loaded as zksnake_amd._pyints: a shadow package directory first on the path carries the sanitized build
clang_c=/opt/rocm/lib/llvm/bin/clang
$clang_c -O1 -shared -fPIC -pthread $SAN -shared-libsan -I"$PYINC" -o "$OUT/_pyints.so" "$CSRC/pyints.c"
$HIPCC -x c++ -O1 -std=c++17 -fPIC -shared -fopenmp=libgomp $SAN -shared-libsan -o "$OUT/libzk_oracle.so" "$ROOT/oracle/zk_oracle.cpp" 2>/dev/null \
  || g++ -O1 -std=c++17 -fPIC -shared -fopenmp -o "$OUT/libzk_oracle.so" "$ROOT/oracle/zk_oracle.cpp"   # oracle unsanitized if clang lacks OpenMP

cd "$ROOT"
export LD_PRELOAD="$CLANG_RT/libclang_rt.asan-x86_64.so"
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1:allocator_may_return_null=1"
export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"
export ZKMI_LIB="$OUT/libzkmi.so" ZKMI_PYINTS="$OUT/_pyints.so" ZK_ORACLE_LIB="$OUT/libzk_oracle.so"
python3 -m pytest -x -q -m "not gpu" -p no:cacheprovider tests/test_host_lib.py tests/test_host_python.py tests/test_plonk_host.py tests/test_oracle.py "$@"

# ThreadSanitizer over the worker pool of the marshalling helper: 2^19 integers (fast path, slow path for values >= 2^256 and
# >= r) converted on eight threads with the chunk callback, three times, result compared with Python's own arithmetic
mkdir -p "$ROOT/build/tsan"
$clang_c -O1 -g -shared -fPIC -pthread -fsanitize=thread -shared-libsan -I"$PYINC" -o "$ROOT/build/tsan/_pyints.so" "$CSRC/pyints.c"
unset ZKMI_LIB ZK_ORACLE_LIB ASAN_OPTIONS UBSAN_OPTIONS
export LD_PRELOAD="$CLANG_RT/libclang_rt.tsan-x86_64.so" TSAN_OPTIONS="halt_on_error=1:report_signal_unsafe=0"
export ZKMI_PYINTS="$ROOT/build/tsan/_pyints.so" ZKMI_PACK_THREADS=8
python3 - <<'PY'
import random
import numpy as np
import zksnake_amd._native as N
from zksnake_amd.constant import BN254_SCALAR_FIELD as r
assert N._pyints.__file__.endswith("build/tsan/_pyints.so")
random.seed(1)
vals = [random.randrange(r) for _ in range(1 << 19)] + [r + 5, 2 ** 300]
for rep in range(3):
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    seen = []
    N.ints_to_limbs(vals, 4, r, out=out, chunk_done=lambda b, e: seen.append((b, e)))
    assert N.limbs_to_ints(out) == [v % r for v in vals] and seen[-1][1] == len(vals)
print("thread sanitizer: worker pool clean")
PY
