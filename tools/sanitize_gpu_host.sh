#!/bin/bash
# UndefinedBehaviorSanitizer over the HOST code of libzkmi.so while it drives the real kernels: the plan bookkeeping, sort
# planning, stage scheduling and MSM tails only run with a GPU behind them, so the CPU pass (tools/sanitize_cpu.sh) cannot see
# them.  The device code is compiled as usual (GPU sanitizers are not available on the pool); UBSan needs no shadow memory,
# so it does not disturb the HIP runtime's address-space layout the way a host AddressSanitizer could.
#   here:           bash tools/sanitize_gpu_host.sh build
#   on the GPU box: gpurun -- 'bash tools/sanitize_gpu_host.sh run > gpurun_out/ubsan_gpu_tests.log 2>&1'
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
RT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.ubsan_standalone-x86_64.so
case "${1:-build}" in
build)
    make -s -j8 -C "$ROOT/zksnake_amd/csrc" OBJDIR=../../build/obj_ubsan LIBOUT=../../build/ubsan/libzkmi.so \
        HOSTSAN="-Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize-recover=undefined -Xarch_host -g" ../../build/ubsan/libzkmi.so
    ;;
run)
    cd "$ROOT"
    export LD_PRELOAD="$RT" UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 ZKMI_LIB="$ROOT/build/ubsan/libzkmi.so"
    timeout -k 10 900 python3 -m pytest tests -m gpu -x -q
    ;;
esac
