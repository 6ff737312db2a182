#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box (run through gpurun):
#   bash tools/collect_profiles.sh <prefix>      e.g. p4  -> gpurun_out/p4_{kt,fetch,write,prove,plonk}
# kernel-trace + stats and the two PMC passes are separate runs (gpurun refuses --pmc combined with tracing domains).
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
P=$1
cd /tmp
export TMPDIR=/tmp
B="$R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${P}_kt -- python3 $B > $R/gpurun_out/${P}_kt.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${P}_fetch -- python3 $B > $R/gpurun_out/${P}_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${P}_write -- python3 $B > $R/gpurun_out/${P}_write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${P}_prove -- python3 $R/tools/prove_bench.py --log-n 20 --reps 4 > $R/gpurun_out/${P}_prove.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${P}_plonk -- python3 $R/tools/plonk_bench.py --log-n 18 --reps 4 > $R/gpurun_out/${P}_plonk.log 2>&1
echo collected
