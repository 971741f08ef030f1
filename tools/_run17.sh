mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t17.log 2>&1; rc=$?; tail -4 gpurun_out/t17.log
