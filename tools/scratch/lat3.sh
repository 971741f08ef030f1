run() { timeout -k 10 200 python bench.py --steps 40 --warmup 5 --cpu-seconds 0 --no-verify --repeats 2 --frames-in-flight 1 "$@" > gpurun_out/bench_w.json 2> gpurun_out/bench_w.err || { tail -5 gpurun_out/bench_w.err; return 1; }; python -c "
import json,sys; d=json.load(open('gpurun_out/bench_w.json')); print(' '.join(sys.argv[1:]), '->', d['ms_per_step'], d['repeat_ms_per_step'], 'dda in frame', d['stages_ms_per_step']['trace_dda'])" "$@"; }
run && run --opt dda_blocks=768 && run --opt dda_blocks=512 && run --opt dda_blocks=1280 && run --opt dda_blocks=1536 && run
