#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) per kernel.

    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r02_pmc_traffic.json [commit]

Units and gfx950 correction per MI355X_MICROARCH.md §HBM: the counters are in KiB; FETCH_SIZE reports half of the
bytes of wide coalesced streaming reads, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact.
"""
import csv, glob, json, re, sys
from collections import defaultdict

def short(name):
    m = re.search(r"(gemm_f32_kernel<[^>]*>|gemm_wk_kernel<[^>]*>|gemm_wk_multi_kernel<[^>]*>|gemm_f32_pair_kernel<[^>]*>|gemm_f32_rms_multi_kernel|conv8_c32_kernel<[^>]*>|[A-Za-z0-9_]+_kernel)", name)
    return m.group(1) if m else name[:60]

def load(d, counter):
    out = defaultdict(list)
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                out[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return out

def summarise(fetch_dir, write_dir, commit="unknown"):
    """{kernel: {launches, read / write / hbm bytes per launch}} + "_gemm_all" (the fp32 MFMA core) + "_meta"."""
    fetch, write = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, []), write.get(k, [])
        if not (k.startswith("gemm") or k.startswith("conv8_c32") or k.endswith("_kernel")) or "at::" in k:
            continue
        rd = 2.0 * 1024 * sum(f) / max(len(f), 1)
        wr = 1024.0 * sum(w) / max(len(w), 1)
        res[k] = dict(launches=len(f), read_bytes_per_launch=rd, write_bytes_per_launch=wr, hbm_bytes_per_launch=rd + wr)
    g = [v for k, v in res.items() if k.startswith("gemm_f32") or k.startswith("gemm_wk") or k.startswith("conv8_c32")]      # the fp32 MFMA core
    n = sum(v["launches"] for v in g)
    res["_gemm_all"] = dict(launches=n, hbm_bytes_per_launch=sum(v["hbm_bytes_per_launch"] * v["launches"] for v in g) / max(n, 1))
    res["_meta"] = dict(launches=0, hbm_bytes_per_launch=0.0, commit=commit,
                        command="rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py --steps 12 "
                                "--warmup 3 --repeats 1 --no-graph --no-cpu-baseline --no-sampling --no-modes --no-extras")
    return res


if __name__ == "__main__":
    res = summarise(sys.argv[1], sys.argv[2], sys.argv[4] if len(sys.argv) > 4 else "unknown")
    json.dump(res, open(sys.argv[3], "w"), indent=1)
    for k, v in res.items():
        print(f"{k:60s} n={v['launches']:5d} hbm/launch={v['hbm_bytes_per_launch']/1e6:8.3f} MB")
