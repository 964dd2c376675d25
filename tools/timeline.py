#!/usr/bin/env python3
"""Per-launch timeline of one group of n_critic iterations from a rocprofv3 --kernel-trace CSV.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-sampling
    python tools/timeline.py gpurun_out/tl [out.txt]

Takes the last complete group (from one group_inputs / gather_batch_slots launch to the next) and prints, per launch: start offset,
duration, gap to the previous launch's end, grid, short name; then the sums per kernel family.
"""
import csv, glob, re, sys
from collections import defaultdict

def short(name):
    m = re.search(r"(gemm_f32_kernel<[^>]*>|gemm_f32_nt128_kernel<[^>]*>|conv8_c32_kernel<[^>]*>|[A-Za-z0-9_]+_kernel(<[^>]*>)?)", name)
    return m.group(1) if m else name[:50]

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]),
                     "x".join(r.get(k, "?") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")),
                     r.get("Workgroup_Size_X", "?")))
rows.sort()
marks = [i for i, r in enumerate(rows) if r[2].startswith("gather_batch") or r[2].startswith("group_inputs")]
if len(marks) < 3:
    sys.exit("no complete group in the trace")
a, b = marks[-3], marks[-2]
grp = rows[a:b]
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
t0, prev_end = grp[0][0], grp[0][0]
fam = defaultdict(lambda: [0, 0.0])
gaps = 0.0
for s, e, n, g, wg in grp:
    gap = (s - prev_end) / 1e3
    gaps += max(gap, 0.0)
    print(f"{(s - t0) / 1e3:9.2f} us  dur {(e - s) / 1e3:7.2f}  gap {gap:6.2f}  grid {g:>16s}/{wg:<4s} {n}", file=out)
    fam[n][0] += 1
    fam[n][1] += (e - s) / 1e3
    prev_end = max(prev_end, e)
span = (rows[b][0] - t0) / 1e3
print(f"\ngroup span {span:.1f} us, {len(grp)} launches, sum of durations {sum(v[1] for v in fam.values()):.1f} us, "
      f"sum of gaps {gaps:.1f} us", file=out)
for n, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"{t:9.1f} us  {c:4d} x {t / c:7.2f}  {n}", file=out)
