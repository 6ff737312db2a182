#!/bin/bash
# Compiles one HIP translation unit like `hipcc -c`, with ONE change in the device pipeline: LLVM's O3 pass pipeline runs without
# its `reassociate` pass.  Why: the field products are written as one accumulator chain per column (field.hip.h); `reassociate`,
# run on the straight-line code that is left after the __forceinline__ layers have been inlined, sorts every column sum by operand
# rank -- products first, the carry of the previous column last -- which starts each column from zero and spends one 64-bit add
# per column on joining the carry: 146 of the 2298 VALU instructions of a BN254 G1 bucket addition.  The
# driver has no switch for a single pass, so the device half goes through its stages by hand:
#   clang (HIP -> unoptimised device bitcode, device libraries linked in) -> opt (the pipeline `opt -O3` prints, minus
#   `reassociate`) -> llc -> lld (code object) -> clang-offload-bundler -> clang host compile with the bundle embedded.
# usage: hipcc_noreassoc.sh <out.o> <source.hip> [compiler flags...]     (ARCH, LLVM_BIN, LLCFLAGS from the environment)
set -euo pipefail
OUT=$1; SRC=$2; shift 2
ARCH=${ARCH:-gfx950}
LLVM=${LLVM_BIN:-/opt/rocm/lib/llvm/bin}
TMP=$(mktemp -d "${TMPDIR:-/tmp}/zknr.XXXXXX")
trap 'rm -rf "$TMP"' EXIT
"$LLVM/clang++" -x hip --offload-arch=$ARCH --cuda-device-only -Xclang -disable-llvm-passes -emit-llvm -c "$SRC" -o "$TMP/dev.bc" "$@"
# (opt prints the pipeline and then fails to re-parse the writer pass it appended itself: the exit status says nothing)
PIPE=$( ("$LLVM/opt" -O3 -print-pipeline-passes "$TMP/dev.bc" -o /dev/null 2>/dev/null || true) | sed 's/,BitcodeWriterPass//')
case "$PIPE" in
    *reassociate,*) ;;
    *)  # another LLVM: the pipeline text cannot be reproduced (or has no such pass).  The stock driver compiles the same sources to
        # the same results, a few per cent slower in the bucket-accumulation kernel: fall back to it instead of failing the build
        # (round-3 advisor finding).  LLCFLAGS travel as -mllvm options.
        echo "hipcc_noreassoc.sh: cannot reproduce the O3 pipeline without 'reassociate' on this toolchain; compiling $SRC with plain hipcc" >&2
        MLLVM=(); for f in ${LLCFLAGS:-}; do MLLVM+=(-mllvm "$f"); done
        exec "${HIPCC:-/opt/rocm/bin/hipcc}" --offload-arch=$ARCH "${MLLVM[@]}" -c "$SRC" -o "$OUT" "$@";;
esac
[ -n "${KEEP_REASSOCIATE:-}" ] || PIPE=${PIPE//reassociate,/}   # KEEP_REASSOCIATE=1: the stock pipeline through the same stages (A/B)
"$LLVM/opt" -mtriple=amdgcn-amd-amdhsa -mcpu=$ARCH -amdgpu-internalize-symbols -passes="$PIPE" "$TMP/dev.bc" -o "$TMP/dev.opt.bc"
# LLCFLAGS: extra code-generation options of one translation unit (the Makefile passes a scheduling strategy for some)
"$LLVM/llc" -mtriple=amdgcn-amd-amdhsa -mcpu=$ARCH -O3 -relocation-model=pic -filetype=obj ${LLCFLAGS:-} "$TMP/dev.opt.bc" -o "$TMP/dev.o"
"$LLVM/lld" -flavor gnu -m elf64_amdgpu --no-undefined -shared -o "$TMP/dev.hsaco" "$TMP/dev.o"
# ZK_KEEP_HSACO=<dir>: keep the code object for tools/isa_stats.py (register counts, instruction mix of a kernel)
[ -z "${ZK_KEEP_HSACO:-}" ] || { mkdir -p "$ZK_KEEP_HSACO"; cp "$TMP/dev.hsaco" "$ZK_KEEP_HSACO/$(basename "$OUT" .o).hsaco"; }
"$LLVM/clang-offload-bundler" -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--$ARCH \
    -input=/dev/null -input="$TMP/dev.hsaco" -output="$TMP/dev.hipfb"
"$LLVM/clang++" -x hip --offload-arch=$ARCH --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang "$TMP/dev.hipfb" -c "$SRC" -o "$OUT" "$@"
