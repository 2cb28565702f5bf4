"""Print the kernel timeline of one tree build from a rocprofv3 --kernel-trace csv (start, duration, grid, kernel)."""
import csv, re, sys
path = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'bvh_init' in r['Kernel_Name']]
a, b = idx[which], idx[which + 1]
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    n = r['Kernel_Name']
    m = re.search(r'(bvh_\w+|tree_walk_wave|walk_\w+|DeviceScan\w*|scan\w*|gather_particles|integrate_inplace|copyBuffer|fillBuffer|radix_sort\w*|merge_sort\w*|transform)', n)
    short = m.group(1) if m else n[:30]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  grid {r.get('Grid_Size_X', '?'):>8} {short}")
