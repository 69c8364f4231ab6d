mkdir -p gpurun_out/r03i
{ echo "# tools/ab_bench.sh inside one gpurun call: kernel arguments in device memory (the package default) vs HIP's default placement"; tools/ab_bench.sh "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0"; } > gpurun_out/r03i/kernarg_ab.txt 2>&1
cat gpurun_out/r03i/kernarg_ab.txt
