mkdir -p gpurun_out
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "radix or full_frame or fused" > gpurun_out/t7.log 2>&1; rc=$?; tail -3 gpurun_out/t7.log
if [ $rc -eq 0 ]; then
for it in 16 8; do echo items $it; timeout -k 10 120 python tools/sort_bench.py --quick --items $it > gpurun_out/sort_bench_i$it.log 2>&1; grep -E '"pairs"' gpurun_out/sort_bench_i$it.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['pairs'], d['key_bits'], d['us'], d['us_per_pass'])"; done
for o in "" "--opt sort_items=8"; do
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 $o > gpurun_out/bench_f.json 2> gpurun_out/bench_f.err; tail -1 gpurun_out/bench_f.err
python -c "
import json; d=json.load(open('gpurun_out/bench_f.json')); print('$o', d['value'], d['ms_per_step'], d['ms_per_step_one_frame_in_flight'], d['verified_against_single_context_frame']); a=d['stages_ms_per_step_alone_on_one_stream']; print('sort stack alone', a['build_sort']+a['sort_rays']+a['shadow_prep'], 'sum', sum(a.values())); b=d['stages_ms_per_step']; print('in frame sort stack', b['build_sort']+b['sort_rays']+b['shadow_prep'], 'sum', sum(b.values()))"
done
fi
