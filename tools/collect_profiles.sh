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
run3 bench $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra
run3 groups $R/tools/group_msm_bench.py 20
run3 ntt $R/tools/ntt_bench.py --log-n 22 --reps 5
run3 prove $R/tools/prove_bench.py --log-n 20 --reps 4
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${P}_plonk_kt -- python3 $R/tools/plonk_bench.py --log-n 18 --reps 4 > $R/gpurun_out/${P}_plonk_kt.log 2>&1
runsq bench $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra
runsq ntt $R/tools/ntt_bench.py --log-n 22 --reps 5
echo collected
