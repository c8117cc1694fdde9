"""Per launch shape (kernel template + grid + workgroup) summary of a rocprofv3 --kernel-trace csv: calls, average us, share of the summed kernel time.
python tools/trace_shapes.py <kernel_trace.csv> [name substring ...]   (filters are OR-ed)"""
import collections
import csv
import re
import sys


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return re.sub(r'\(.*$', '', n)[:70]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    filt = sys.argv[2:]
    agg = collections.defaultdict(list)
    total = 0.0
    for r in rows:
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        total += d
        name = short(r['Kernel_Name'])
        if filt and not any(f in name for f in filt):
            continue
        grid = tuple(int(r.get(k, 0) or 0) for k in ('Grid_Size_X', 'Grid_Size_Y', 'Grid_Size_Z'))
        wg = tuple(int(r.get(k, 0) or 0) for k in ('Workgroup_Size_X', 'Workgroup_Size_Y', 'Workgroup_Size_Z'))
        wgs = tuple(g // max(w, 1) for g, w in zip(grid, wg))
        agg[(name, wgs, wg[0])].append(d)
    print(f'summed kernel time {total / 1e3:.2f} ms over {len(rows)} dispatches')
    for (name, wgs, w), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f'{sum(v) / total * 100:5.2f}%  n={len(v):4d}  avg {sum(v) / len(v):8.1f} us  min {min(v):8.1f}  wgs {wgs} x {w:4d}  {name}')


if __name__ == '__main__':
    main()
