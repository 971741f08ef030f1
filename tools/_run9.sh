mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bounce or reflection or soups or full_size or all_rays_miss or degenerate" > gpurun_out/t9.log 2>&1; rc=$?; tail -3 gpurun_out/t9.log
if [ $rc -eq 0 ]; then
timeout -k 10 200 python tools/dda_sweep.py --quick --out gpurun_out/dda_sweep_r03d.json > gpurun_out/sweep9.log 2>&1; grep -v sharing gpurun_out/sweep9.log | tail -22
fi
