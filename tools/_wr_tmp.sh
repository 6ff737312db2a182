cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/wr_prof -o wr -- python3 $GRAFT_REPO_ROOT/tools/window_range_bench.py 1 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; cat gpurun_out/wr.log | grep -v amdgpu; python3 - <<'P'
import csv,glob
f=glob.glob('gpurun_out/wr_prof/**/*kernel_stats.csv', recursive=True)
rows=list(csv.DictReader(open(f[0])))
for r in rows[:24]:
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(5), f"{float(r['AverageNs'])/1e3:10.1f}")
P
