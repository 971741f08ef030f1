mkdir -p gpurun_out
run() { name=$1; shift; timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 "$@" > gpurun_out/cfg3_$name.json 2> gpurun_out/cfg3_$name.err; python -c "
import json,sys; d=json.load(open('gpurun_out/cfg3_$name.json')); print('$name', d['value'], 'Mrays/s', d['ms_per_step'], 'ms', 'latency', d['ms_per_step_one_frame_in_flight'], 'verified', d['verified_against_single_context_frame'], 'roofline', d['roofline']['kernel'], d['roofline']['frac'], 'fps', d['config']['frames_per_s'])"; }
run hall --workload hall --no-reflect --width 1024 --height 1024
run 4k --width 3840 --height 2160
run anim --animate
run onestream --no-overlap --frames-in-flight 1 --waiting-builds
UGRT_BENCH_REHEARSE=1 timeout -k 10 400 python bench.py --gpus 2 --steps 6 --warmup 2 --cpu-seconds 0 --repeats 0 --scale 0.2 --width 1280 --height 720 > gpurun_out/cfg3_reh2.json 2> gpurun_out/cfg3_reh2.err; python -c "
import json; d=json.load(open('gpurun_out/cfg3_reh2.json')); print('rehearsal 2 ranks', d['n_gpus'], d['verified_against_single_context_frame'], d['band_bounds_tile_rows'])"
