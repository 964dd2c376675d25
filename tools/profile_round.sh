#!/bin/bash
# The round's profile set on the GPU box: kernel stats of the bench command, the two PMC traffic passes, a timeline.
#   bash tools/profile_round.sh <tag> <commit>
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
COMMIT=${2:-unknown}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$TAG $R/gpurun_out/pmc_fetch_$TAG $R/gpurun_out/pmc_write_$TAG
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 60 --warmup 6 --repeats 1 --no-cpu-baseline --no-sampling --no-modes --no-extras > $R/gpurun_out/bench_prof_$TAG.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch_$TAG -- python3 $R/bench.py --steps 12 --warmup 3 --repeats 1 --no-graph --no-cpu-baseline --no-sampling --no-modes --no-extras > $R/gpurun_out/pmc_fetch_$TAG.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write_$TAG -- python3 $R/bench.py --steps 12 --warmup 3 --repeats 1 --no-graph --no-cpu-baseline --no-sampling --no-modes --no-extras > $R/gpurun_out/pmc_write_$TAG.log 2>&1
cd $R
python3 tools/pmc_traffic.py gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG gpurun_out/pmc_traffic_$TAG.json $COMMIT > gpurun_out/pmc_traffic_$TAG.txt
cp $(ls gpurun_out/prof_$TAG/*/*kernel_stats.csv | head -1) gpurun_out/kernel_stats_$TAG.csv
python3 tools/timeline.py gpurun_out/prof_$TAG gpurun_out/timeline_$TAG.txt || true
# keep only the summaries (the raw traces are large)
rm -rf gpurun_out/prof_$TAG gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG
