"""Idle time between consecutive kernels of the resident-batch pipeline under hipGraph replay, from a rocprofv3 kernel trace:
  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -o gaps -- python3 bench.py --no-profile --no-host-loop --no-b1 --cpu-sample 0 --steps 5
  python tools/graph_gaps.py <dir>
Reports, for the last 3 steps: busy time, idle time, and the largest gaps with the kernels on either side."""
import csv, glob, os, sys
p = sorted(glob.glob(os.path.join(sys.argv[1], "**", "gaps*kernel_trace.csv"), recursive=True))[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]) for r in csv.DictReader(open(p))), key=lambda t: t[0])
# the last ~3 steps: take the final 3300 kernels
rows = rows[-3300:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy, gaps, cur_end = 0, [], rows[0][1]
busy += rows[0][1] - rows[0][0]
for (s, e, n), prev in zip(rows[1:], rows[:-1]):
    if s > cur_end:
        gaps.append((s - cur_end, prev[2], n))
        busy += e - s
    else:
        busy += max(0, e - cur_end)
    cur_end = max(cur_end, e)
span = t1 - t0
print(f"span {span/1e6:.3f} ms  busy {busy/1e6:.3f} ms  idle {(span-busy)/1e6:.3f} ms over {len(rows)} kernels; gaps: n={len(gaps)} mean {sum(g[0] for g in gaps)/max(len(gaps),1)/1e3:.2f} us")
import collections
hist = collections.Counter(min(int(g[0] / 500), 20) for g in gaps)
print("gap histogram (0.5 us bins):", sorted(hist.items()))
for g in sorted(gaps, reverse=True)[:12]:
    print(f"  {g[0]/1e3:8.1f} us  after {g[1]}  before {g[2]}")
