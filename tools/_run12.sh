mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t12.log 2>&1; rc=$?; tail -4 gpurun_out/t12.log
if [ $rc -eq 0 ]; then
for o in "" "--opt primary_order=0"; do
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 $o > gpurun_out/bench_g.json 2> gpurun_out/bench_g.err; tail -1 gpurun_out/bench_g.err
python -c "
import json; d=json.load(open('gpurun_out/bench_g.json')); print('$o', d['value'], d['ms_per_step'], d['ms_per_step_one_frame_in_flight'], d['verified_against_single_context_frame'], d['roofline']); a=d['stages_ms_per_step_alone_on_one_stream']; print(a); print('sum', sum(a.values())); b=d['stages_ms_per_step']; print(b); print('sum', sum(b.values()))"
done
fi
