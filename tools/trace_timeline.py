"""Print the kernel timeline of one tree step from a rocprofv3 --kernel-trace csv: start, duration, idle gap before the
kernel, grid, kernel; then the step's length, the sum of its gaps and the largest gap.
    python tools/trace_timeline.py <kernel_trace.csv> [which step = 3] [first kernel of a step = bvh_init]
"""
import csv, re, sys
path = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 3
first = sys.argv[3] if len(sys.argv) > 3 else 'bvh_init'
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if first in r['Kernel_Name']]
a, b = idx[which], idx[which + 1]
t0 = int(rows[a]['Start_Timestamp'])
prev_end = None
gaps = []
for r in rows[a:b]:
    n = r['Kernel_Name']
    m = re.search(r'(bvh_\w+|quad_\w+|qb_\w+|tree_walk_\w+|walk_\w+|DeviceScan\w*|scan\w*|gather_particles|integrate_inplace|copyBuffer|fillBuffer|radix_sort\w*|onesweep\w*|merge_sort\w*|transform)', n)
    short = m.group(1) if m else n[:30]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    gaps.append((gap, short))
    prev_end = max(e, prev_end or e)
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  gap {gap:6.1f} us  grid {r.get('Grid_Size_X', '?'):>8} {short}")
nxt = int(rows[b]['Start_Timestamp'])
print(f"step: {(nxt - t0) / 1e3:.1f} us from its first kernel to the next step's first kernel; "
      f"{len(gaps)} launches; idle inside the step {sum(g for g, _ in gaps if g > 0):.1f} us; "
      f"largest gap {max(gaps)[0]:.1f} us (before {max(gaps)[1]}); gap to the next step {(nxt - prev_end) / 1e3:.1f} us")
