#!/bin/bash
# SQ counter passes for one DDA kernel variant: tools/pmc_sq.sh TAG KERNEL RPW   (run on the GPU box)
set -e
TAG=$1; K=$2; RPW=$3
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/p1 -o p1 --output-format csv -- python3 tools/dda_only.py $K $RPW 3 > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM -d $OUT/p2 -o p2 --output-format csv -- python3 tools/dda_only.py $K $RPW 3 > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/p3 -o p3 --output-format csv -- python3 tools/dda_only.py $K $RPW 3 > $OUT/p3.log 2>&1 || true
find $OUT -name "*counter_collection.csv" | head
