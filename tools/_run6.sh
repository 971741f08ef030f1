mkdir -p gpurun_out/prof6; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof6 -o s --output-format csv -- python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --repeats 0 --no-verify > gpurun_out/prof6/bench.json 2> gpurun_out/prof6/err.log
ls gpurun_out/prof6/*; python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof6/*/*kernel_stats.csv')+glob.glob('gpurun_out/prof6/*kernel_stats.csv')
rows=list(csv.DictReader(open(f[0])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:45]:
    print('%-60s %6s %9.1f us avg  %5.1f%%' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
print('total calls', sum(int(r['Calls']) for r in rows))
PY
