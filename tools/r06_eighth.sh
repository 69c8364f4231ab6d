#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/decode_latency.py > gpurun_out/r06_g_decode_latency.txt 2>&1; grep -v amdgpu.ids gpurun_out/r06_g_decode_latency.txt | tail -12
