cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r03_a > gpurun_out/profile_r03_a.log 2>&1; tail -3 gpurun_out/profile_r03_a.log
python tools/pmc_traffic.py gpurun_out/prof_r03_a r03_a > gpurun_out/pmc_fold.log 2>&1; tail -16 gpurun_out/pmc_fold.log
mkdir -p gpurun_out/profiles_out; cp profiles/r03_a_* profiles/traffic.json gpurun_out/profiles_out/
