run() { timeout -k 10 250 python bench.py --steps 40 --warmup 5 --cpu-seconds 0 --no-verify --repeats 2 "$@" > gpurun_out/bench_w.json 2> gpurun_out/bench_w.err || { tail -5 gpurun_out/bench_w.err; return 1; }; python -c "
import json,sys; d=json.load(open('gpurun_out/bench_w.json')); print(' '.join(sys.argv[1:]), '->', d['ms_per_step'], d['repeat_ms_per_step'])" "$@"; }
run && run --frames-in-flight 3 && run --frames-in-flight 5 && run --frames-in-flight 6 && run --frames-in-flight 8 && run --frames-in-flight 2 && run
