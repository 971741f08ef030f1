run() { timeout -k 10 250 python bench.py --steps 30 --warmup 5 --cpu-seconds 0 --no-verify --repeats 2 "$@" > gpurun_out/bench_w.json 2> gpurun_out/bench_w.err || { tail -5 gpurun_out/bench_w.err; return 1; }; python -c "
import json,sys; d=json.load(open('gpurun_out/bench_w.json')); print(' '.join(sys.argv[1:]), '->', d['ms_per_step'], d['repeat_ms_per_step'], d['ms_per_step_one_frame_in_flight'], 'primary in frame', d['stages_ms_per_step']['trace_primary'], 'alone', d['stages_ms_per_step_alone_on_one_stream']['trace_primary'])" "$@"; }
H="--workload hall --no-reflect --width 1024 --height 1024"
for r in 256 512 1024 2048; do run --opt primary_xcd_run=$r; run $H --opt primary_xcd_run=$r; done
run --opt primary_waves=16384; run $H --opt primary_waves=16384
