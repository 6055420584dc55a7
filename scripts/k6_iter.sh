#!/bin/bash
# development loop of the stage-machine kernel on the GPU box: parity at C3 / C5 / small runs (both forms), then timing
mkdir -p gpurun_out/k6
timeout -k 10 600 python -m pytest tests/test_gpu_sampler_size.py -q -m gpu --timeout 300 -x > gpurun_out/k6/tests.log 2>&1; rc=$?
tail -2 gpurun_out/k6/tests.log
[ $rc -ne 0 ] && { grep -E "^E  |Error" gpurun_out/k6/tests.log | head -8; exit $rc; }
SWEEP_SPLIT=1 WALKERS=${WALKERS:-128} STEPS=4 timeout -k 10 300 python scripts/sampler_bench.py 2>&1 | grep "W="
if [ -f pathintegralgroundstate_amd/libpigs_hip_timing.so ]; then
  PIGS_LIB=$PWD/pathintegralgroundstate_amd/libpigs_hip_timing.so SWEEP_SPLIT=1 TIMING=1 WALKERS=128 STEPS=3 timeout -k 10 300 python scripts/sampler_bench.py 2>&1 | grep "shader-clock"
fi
