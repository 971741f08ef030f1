#!/bin/bash
# Launch-shape sweep of the default bench frame (4 frames in flight): which settings of bench.py's constants still hold
# after a round's kernel changes?   tools/throughput_sweep.sh > gpurun_out/throughput_sweep.txt
B="python3 bench.py --cpu-seconds 0 --no-other-configs --no-verify --repeats 2 --steps 20 --warmup 5"
run() { tag=$1; shift; out=$("$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['repeat_ms_per_step'], d['ms_per_step_one_frame_in_flight'], d['stages_ms_per_step'].get('trace_dda'))"); echo "$tag: $out"; }
run default $B
run dda_split=1 $B --opt dda_split=1
run dda_rpw=32 $B --opt dda_rays_per_wave=32
run dda_blocks=2048 $B --opt dda_blocks=2048
run fif=2 $B --frames-in-flight 2
run fif=3 $B --frames-in-flight 3
run fif=5 $B --frames-in-flight 5
run fif=6 $B --frames-in-flight 6
run fif=8 $B --frames-in-flight 8
GPU_MAX_HW_QUEUES=8 run queues=8 $B
GPU_MAX_HW_QUEUES=6 run queues=6 $B
GPU_MAX_HW_QUEUES=2 run queues=2 $B
run sort_items=8 $B --opt sort_items=8
run sort_rank=0 $B --opt sort_rank=0
run shadow_waves=4096 $B --opt shadow_waves=4096
