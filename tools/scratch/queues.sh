run() { q=$1; shift; GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python bench.py --steps 30 --warmup 5 --cpu-seconds 0 --no-verify --repeats 2 "$@" > gpurun_out/bench_q.json 2> gpurun_out/bench_q.err; python -c "
import json,sys; d=json.load(open('gpurun_out/bench_q.json')); print('queues', sys.argv[1], ' '.join(sys.argv[2:]), '->', d['ms_per_step'], d['repeat_ms_per_step'])" $q "$@"; }
run 4
run 8
run 6
run 2
run 4 --frames-in-flight 2
run 8 --frames-in-flight 8
run 16 --frames-in-flight 8
