mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/t4.log 2>&1; rc=$?; tail -5 gpurun_out/t4.log
if [ $rc -eq 0 ]; then
timeout -k 10 120 python tools/sort_bench.py > gpurun_out/sort_bench_d.log 2>&1; grep -E '"pairs": (131072|1048576|2097152|8388608), "key_bits": (16|32), "keys": "random"' gpurun_out/sort_bench_d.log
for o in "" "--opt sort_fused_hist=0"; do
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 $o > gpurun_out/bench_d.json 2> gpurun_out/bench_d.err; tail -2 gpurun_out/bench_d.err
python -c "
import json; d=json.load(open('gpurun_out/bench_d.json')); print('$o', d['value'], d['ms_per_step'], d['ms_per_step_one_frame_in_flight'], d['verified_against_single_context_frame'], d['roofline']['frac']); a=d['stages_ms_per_step_alone_on_one_stream']; print(a); print('sort stack alone', a['build_sort']+a['sort_rays']+a['shadow_prep'], 'sum', sum(a.values())); print(d['stages_ms_per_step'])"
done
fi
