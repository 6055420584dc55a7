#!/bin/bash
# development loop of the sampler kernels on the GPU box: parity at C3 / C5 / small runs (all forms), then timing
mkdir -p gpurun_out/k6
timeout -k 10 900 python -m pytest tests/test_gpu_sampler_size.py -q -m gpu --timeout 300 -x ${K6_TESTS:+-k "$K6_TESTS"} > gpurun_out/k6/tests.log 2>&1; rc=$?
tail -2 gpurun_out/k6/tests.log
[ $rc -ne 0 ] && { grep -E "^E  |Error" gpurun_out/k6/tests.log | head -12; exit $rc; }
for cm in 0 -1; do
  echo "cm_split=$cm"
  CM_SPLIT=$cm SWEEP_SPLIT=${SWEEP_SPLIT:-0} WALKERS=${WALKERS:-128} STEPS=4 timeout -k 10 300 python scripts/sampler_bench.py 2>&1 | grep "W="
done
if [ -n "$K6_TIMING" ] && [ -f pathintegralgroundstate_amd/libpigs_hip_timing.so ]; then
  PIGS_LIB=$PWD/pathintegralgroundstate_amd/libpigs_hip_timing.so SWEEP_SPLIT=1 TIMING=1 WALKERS=128 STEPS=3 timeout -k 10 300 python scripts/sampler_bench.py 2>&1 | grep "shader-clock"
fi
