cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r03_d > gpurun_out/profile_r03_d.log 2>&1; tail -2 gpurun_out/profile_r03_d.log
python tools/pmc_traffic.py gpurun_out/prof_r03_d r03_d > gpurun_out/pmc_fold_d.log 2>&1; tail -12 gpurun_out/pmc_fold_d.log
mkdir -p gpurun_out/profiles_out_d; cp profiles/r03_d_* profiles/traffic.json gpurun_out/profiles_out_d/
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_final_r03.json 2> gpurun_out/bench_final_r03.err; tail -2 gpurun_out/bench_final_r03.err; python -c "
import json; d=json.load(open('gpurun_out/bench_final_r03.json')); print(d['value'], d['ms_per_step'], d['ms_per_step_one_frame_in_flight'], d['verified_against_single_context_frame'], d['roofline']['frac'], d['roofline']['traffic'], d['frame_hbm'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
