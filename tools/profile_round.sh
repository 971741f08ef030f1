#!/bin/bash
# One round's profile evidence, on the GPU box:  tools/profile_round.sh TAG [bench flags...]
#   gpurun_out/prof_TAG/stats   rocprofv3 --kernel-trace --stats of the default bench command
#   gpurun_out/prof_TAG/fetch   --pmc FETCH_SIZE            (separate passes, as MI355X_MICROARCH.md prescribes)
#   gpurun_out/prof_TAG/write   --pmc WRITE_SIZE
#   gpurun_out/prof_TAG/sq      --pmc SQ_* (issue / wait / occupancy of every kernel)
#   gpurun_out/prof_TAG/kernel_resources.txt   registers / spills / scratch / LDS of every kernel (tools/kernel_resources.py)
# Fold them afterwards with tools/pmc_traffic.py into profiles/.
# The first run is the default command (its line also carries the other BASELINE configs and the verification).  The
# passes under rocprofv3 add --no-verify --no-other-configs: their per-kernel averages are then those of the timed
# configuration alone (the verifying context runs one stream, other launch shapes and waiting builds; the other configs
# other sizes), which is what roofline.traffic / frame_hbm are attributed to.
set -e
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
python3 tools/kernel_resources.py -o $OUT/kernel_resources.txt > /dev/null
python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 "$@" > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --repeats 0 --no-verify --no-other-configs "$@" > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --repeats 0 --no-verify --no-other-configs "$@" > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --repeats 0 --no-verify --no-other-configs "$@" > /dev/null 2> $OUT/write.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/sq -o q --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --repeats 0 --no-verify --no-other-configs "$@" > /dev/null 2> $OUT/sq.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM -d $OUT/sq2 -o q2 --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --repeats 0 --no-verify --no-other-configs "$@" > /dev/null 2> $OUT/sq2.err
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/l2 -o l --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --repeats 0 --no-verify --no-other-configs "$@" > /dev/null 2> $OUT/l2.err || true
rm -f $OUT/*/*agent_info.csv
ls $OUT/*
