mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "primary or full_frame or depth_layers or split_cells or wide or soups or z_slabs" > gpurun_out/t10.log 2>&1; rc=$?; tail -3 gpurun_out/t10.log
if [ $rc -eq 0 ]; then
timeout -k 10 300 python tools/primary_sweep.py --out gpurun_out/primary_sweep_r03.json > gpurun_out/psweep.log 2>&1; tail -16 gpurun_out/psweep.log
fi
