#!/bin/bash
# Per-kernel average durations of the bench step under rocprofv3 --kernel-trace --stats (in-step timings, graph replay).
#   bash tools/kstat.sh <tag> [ENV=val ...]      -> gpurun_out/kstat_<tag>.csv + the top rows on stdout
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do export "$e"; done
rm -rf $R/gpurun_out/kprof_$TAG
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kprof_$TAG -- python3 $R/bench.py --steps 60 --warmup 6 --repeats 1 --no-cpu-baseline --no-sampling --no-modes --no-extras > $R/gpurun_out/kstat_$TAG.log 2>&1 || { tail -5 $R/gpurun_out/kstat_$TAG.log; exit 1; }
cp $(ls $R/gpurun_out/kprof_$TAG/*/*kernel_stats.csv | head -1) $R/gpurun_out/kstat_$TAG.csv
rm -rf $R/gpurun_out/kprof_$TAG
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/kstat_$TAG.csv")))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("$TAG total kernel ms %.2f" % (tot / 1e6))
for r in rows[:${TOP:-8}]:
    print("%8.2f us x %5s  %5.1f%%  %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], 100 * float(r["TotalDurationNs"]) / tot, r["Name"][:90]))
PY
