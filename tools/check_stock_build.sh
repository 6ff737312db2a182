#!/bin/bash
# The default build runs the device code through LLVM's O3 pipeline WITHOUT its `reassociate` pass (csrc/hipcc_noreassoc.sh).  This
# check pins that build against stock hipcc output (round-3 advisor finding): build the library a second time with NOREASSOC=0,
# run the parity tests through it (ZKMI_LIB) and the headline bench through both.
#   here (no GPU):   bash tools/check_stock_build.sh build
#   on the GPU box:  bash tools/check_stock_build.sh run      (through gpurun; writes gpurun_out/stock_*)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
case "${1:-build}" in
build)
    make -C $R/zksnake_amd/csrc -j6 ARCH=gfx950 NOREASSOC=0 OBJDIR=../../build/obj_stock LIBOUT=../../build/variants/libzkmi_stock.so ../../build/variants/libzkmi_stock.so
    ;;
run)
    cd $R
    ZKMI_LIB=$R/build/variants/libzkmi_stock.so timeout -k 10 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_ntt.py tests/test_gpu_codec.py \
        "tests/test_gpu_groth16.py::test_proof_bytes_equal_closed_form" "tests/test_gpu_groth16.py::test_full_size_proof_equals_committed_closed_form" \
        tests/test_gpu_plonk.py -x -q -m gpu > gpurun_out/stock_pytest.log 2>&1
    timeout -k 10 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > gpurun_out/stock_bench_default_build.json 2>/dev/null
    ZKMI_LIB=$R/build/variants/libzkmi_stock.so timeout -k 10 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > gpurun_out/stock_bench_stock_build.json 2>/dev/null
    tail -1 gpurun_out/stock_pytest.log
    ;;
esac
