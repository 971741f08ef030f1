mkdir -p gpurun_out
run() { timeout -k 10 200 python bench.py --steps 30 --warmup 5 --cpu-seconds 0 --no-verify --repeats 2 "$@" > gpurun_out/bench_h.json 2> gpurun_out/bench_h.err; python -c "
import json,sys; d=json.load(open('gpurun_out/bench_h.json')); print(' '.join(sys.argv[1:]), '->', d['ms_per_step'], d['repeat_ms_per_step'], 'dda in-frame', d['stages_ms_per_step']['trace_dda'])" "$@"; }
run
run --frames-in-flight 6
run --frames-in-flight 3
run --opt dda_blocks=4096
run --opt dda_blocks=2048
run --opt dda_rays_per_wave=64
run --opt dda_rays_per_wave=16
run --opt shadow_waves=8192
run --opt primary_waves=8192
run --opt primary_chunk=16
