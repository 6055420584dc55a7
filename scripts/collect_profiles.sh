#!/bin/bash
# On the GPU box: the bench line, the rocprofv3 kernel statistics of the same command and the two PMC passes for K1's
# memory-side traffic.  Outputs under gpurun_out/<tag>/; scripts/collect_profiles.py condenses them into profiles/.
#   usage: scripts/collect_profiles.sh r02
set -o pipefail
tag=${1:-r03}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
python3 bench.py > $out/bench_line.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
tail -c 600 $out/bench_line.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 200 --warmup 10 --no-cpu --mc-steps 3 --mc-large-walkers 0 > $out/stats.log 2>&1 || { tail -5 $out/stats.log; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 20 --warmup 2 --no-cpu --no-mc > $out/fetch.log 2>&1 || { tail -5 $out/fetch.log; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 20 --warmup 2 --no-cpu --no-mc > $out/write.log 2>&1 || { tail -5 $out/write.log; exit 1; }
find $out -name "*kernel_stats.csv" | head -3
