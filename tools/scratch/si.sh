run() { timeout -k 10 250 python bench.py --steps 40 --warmup 5 --cpu-seconds 0 --no-verify --repeats 2 "$@" > gpurun_out/bench_w.json 2> gpurun_out/bench_w.err || { tail -5 gpurun_out/bench_w.err; return 1; }; python -c "
import json,sys; d=json.load(open('gpurun_out/bench_w.json')); f=d['stages_ms_per_step']; print(' '.join(sys.argv[1:]), '->', d['ms_per_step'], d['repeat_ms_per_step'], d['ms_per_step_one_frame_in_flight'], 'sorts in frame', f['build_sort'], f['sort_rays'], f['shadow_prep'])" "$@"; }
run && run --opt sort_items=8 && run --opt sort_items=16 && run
