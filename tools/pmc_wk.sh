#!/bin/bash
# PMC passes (cache hit / fabric traffic) for the wk kernel and the slab kernel on the critic's first layer
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in new old; do
  extra=""; [ $v = old ] && extra="old"
  for pmc in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    d=$R/gpurun_out/pmc_wk/${v}_$(echo $pmc | tr ' ' '_')
    timeout -s KILL 90 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $d -- python3 $R/tools/wk_one.py 0 128 2048 1501 1504 1504 1 6 $extra > /dev/null 2>&1 || echo "pass failed: $v $pmc"
  done
done
python3 - <<'PY'
import csv, glob, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_wk"
for d in sorted(glob.glob(root + "/*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gemm" in k:
                acc[(k[:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in acc.items():
            print(os.path.basename(d), k, c, "avg/launch", sum(v) / len(v), "n", len(v))
PY
