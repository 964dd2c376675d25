#!/usr/bin/env python3
"""Wave-time breakdown of the contraction kernels in one rocprofv3 --pmc pass (with --kernel-trace).

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU \
        SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_x \
        -- python tools/gemm_one.py <layout M N K lda ldb ns reps mode>
    python tools/pmc_breakdown.py gpurun_out/pmc_x "label"
"""
import collections, csv, glob, json, sys
d_, label = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else sys.argv[1])
acc, dur = collections.defaultdict(list), []
for f in glob.glob(d_ + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_" in r["Kernel_Name"] or "conv8_" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(d_ + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_" in r["Kernel_Name"] or "conv8_" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
d = {k: sum(v[3:]) / len(v[3:]) for k, v in acc.items()}
cyc = d["GRBM_GUI_ACTIVE"] / 8
wc = d["SQ_WAVE_CYCLES"]
print(json.dumps({label: dict(avg_us=sum(dur[3:]) / len(dur[3:]), kernel_cycles=cyc,
                              mfma_busy_frac=d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc),
                              wave_wait_any=d["SQ_WAIT_ANY"] / wc, wave_wait_inst=d["SQ_WAIT_INST_ANY"] / wc,
                              wave_active=d["SQ_ACTIVE_INST_ANY"] / wc, valu_insts=d["SQ_INSTS_VALU"],
                              lds_conflict_frac=d["SQ_LDS_BANK_CONFLICT"] / max(d["SQ_LDS_IDX_ACTIVE"], 1))}))
