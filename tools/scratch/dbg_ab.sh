#!/bin/bash
# correctness of product and crosslane variant at 2^21/2^22, then timing of both (sizes 21 22)
LIB=plonky2_goldibear_amd/lib/libgoldibear_gpu.so
timeout -k 10 500 python tools/scratch/dbg_large.py > gpurun_out/dbg_product.log 2>&1
cp $LIB /tmp/product.so; cp tools/bin/libs/crosslane.so $LIB
timeout -k 10 500 python tools/scratch/dbg_large.py > gpurun_out/dbg_crosslane.log 2>&1
cp /tmp/product.so $LIB
GB_LS_SIZES="21 22" bash tools/large_sizes.sh product crosslane > gpurun_out/ls.log 2>&1
