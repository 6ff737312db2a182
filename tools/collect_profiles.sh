#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box (run through gpurun):
#   bash tools/collect_profiles.sh <prefix>      e.g. p4  -> gpurun_out/p4_<workload>_{kt,fetch,write}
# kernel-trace + stats and the two PMC passes are separate runs (gpurun refuses --pmc combined with tracing domains);
# the program itself follows `--` (no env / bash -c hop under the profiler).
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
P=$1
cd /tmp
export TMPDIR=/tmp
run3() {  # name, program args...
    local name=$1; shift
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${P}_${name}_kt -- python3 "$@" > $R/gpurun_out/${P}_${name}_kt.log 2>&1
    timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${P}_${name}_fetch -- python3 "$@" > $R/gpurun_out/${P}_${name}_fetch.log 2>&1
    timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${P}_${name}_write -- python3 "$@" > $R/gpurun_out/${P}_${name}_write.log 2>&1
    echo "$name done"
}
runsq() {  # name, program args...: SQ issue / wait counters of every kernel (one pass; eight SQ counters fit together)
    local name=$1; shift
    timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/${P}_${name}_sq -- python3 "$@" > $R/gpurun_out/${P}_${name}_sq.log 2>&1
    echo "$name sq done"
}
runlds() {  # name, program args...: LDS bank-conflict / activity counters (their own pass: the SQ pass above is full)
    local name=$1; shift
    timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/${P}_${name}_lds -- python3 "$@" > $R/gpurun_out/${P}_${name}_lds.log 2>&1
    echo "$name lds done"
}
# the bench passes run the DRIVER's command, so that the kernel averages under profiles/ are those of the line the driver records
# (round-3 verdict: a four-launch run on cold clocks was 13-15 % off)
BENCH_ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-extra"
run3 bench $R/bench.py $BENCH_ARGS
run3 groups $R/tools/group_msm_bench.py 20
run3 ntt $R/tools/ntt_bench.py --log-n 22 --reps 5
run3 prove $R/tools/prove_bench.py --log-n 20 --reps 4
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${P}_plonk_kt -- python3 $R/tools/plonk_bench.py --log-n 18 --reps 4 > $R/gpurun_out/${P}_plonk_kt.log 2>&1
runsq bench $R/bench.py $BENCH_ARGS
runsq ntt $R/tools/ntt_bench.py --log-n 22 --reps 5
runlds ntt $R/tools/ntt_bench.py --log-n 22 --reps 5
echo collected
