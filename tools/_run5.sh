mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/t5.log 2>&1; rc=$?; tail -5 gpurun_out/t5.log
if [ $rc -eq 0 ]; then
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > gpurun_out/bench_e.json 2> gpurun_out/bench_e.err; tail -2 gpurun_out/bench_e.err
python -c "
import json; d=json.load(open('gpurun_out/bench_e.json')); print(d['value'], d['ms_per_step'], d['ms_per_step_one_frame_in_flight'], d['verified_against_single_context_frame'], d['roofline']['frac']); a=d['stages_ms_per_step_alone_on_one_stream']; print(a); print('sort stack alone', a['build_sort']+a['sort_rays']+a['shadow_prep'], 'sum', sum(a.values())); print(d['stages_ms_per_step'])"
timeout -k 10 300 python tools/shard_cost.py --out gpurun_out/shard_cost.json > gpurun_out/shard_cost.log 2>&1; tail -30 gpurun_out/shard_cost.log
fi
