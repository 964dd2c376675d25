#!/usr/bin/env python3
"""MFMA-pipe utilisation of the contraction kernels from a rocprofv3 --pmc pass.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
        -d gpurun_out/pmc_mfma_m1 -- python tools/gemm_one.py 0 8192 2048 1024 1024 1024 1 20 1
    python tools/pmc_mfma.py gpurun_out/pmc_mfma_m1 [label]

SQ_VALU_MFMA_BUSY_CYCLES counts the cycles the matrix pipes are busy, summed over the chip's 1024 SIMDs (MI355X_MICROARCH.md
cycle-constants table: 32 per v_mfma_f32_32x32x16_bf16, 64 per v_mfma_f32_32x32x2_f32); GRBM_GUI_ACTIVE is summed over
the 8 XCDs, so kernel cycles = GRBM_GUI_ACTIVE / 8 and  MFMA busy = BUSY / (1024 * kernel cycles)."""
import csv, glob, json, re, sys
from collections import defaultdict

d = sys.argv[1]
label = sys.argv[2] if len(sys.argv) > 2 else d
vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(gemm_\w+_kernel|conv8_\w+_kernel)", r["Kernel_Name"])
        if m:
            vals[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(gemm_\w+_kernel|conv8_\w+_kernel)", r["Kernel_Name"])
        if m:
            dur[m.group(1)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {}
for k, c in vals.items():
    n = len(c["GRBM_GUI_ACTIVE"])
    skip = min(3, n - 1)                       # first launches: cold caches / clock ramp
    avg = lambda name: sum(c[name][skip:]) / max(len(c[name][skip:]), 1)
    cyc = avg("GRBM_GUI_ACTIVE") / 8.0
    busy = avg("SQ_VALU_MFMA_BUSY_CYCLES")
    us = sum(dur[k][skip:]) / max(len(dur[k][skip:]), 1) if dur[k] else None
    out[k] = dict(launches=n, kernel_cycles=cyc, mfma_busy_cycles_all_simds=busy, mfma_busy_frac=busy / (1024.0 * cyc),
                  avg_us=us, clock_ghz=(cyc / us / 1e3) if us else None)
print(json.dumps({label: out}))
