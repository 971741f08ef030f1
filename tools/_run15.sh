mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t15.log 2>&1; rc=$?; tail -4 gpurun_out/t15.log
if [ $rc -eq 0 ]; then
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > gpurun_out/bench_i.json 2> gpurun_out/bench_i.err; tail -1 gpurun_out/bench_i.err
python -c "
import json; d=json.load(open('gpurun_out/bench_i.json')); print(d['value'], d['ms_per_step'], d['ms_per_step_one_frame_in_flight'], d['verified_against_single_context_frame'], d['roofline']['frac'], d['roofline']['traffic']); a=d['stages_ms_per_step_alone_on_one_stream']; print(a); print('sum', sum(a.values()))"
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof15 -o s --output-format csv -- python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --repeats 0 --no-verify > gpurun_out/prof15_bench.json 2> gpurun_out/prof15_err.log
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof15/*kernel_stats.csv')
rows=list(csv.DictReader(open(f[0])))
print('total calls', sum(int(r['Calls']) for r in rows), 'k_trace_primary calls', [r['Calls'] for r in rows if 'k_trace_primary<true, false>' in r['Name']])
PY
fi
