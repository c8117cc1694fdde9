"""One train step's kernel timeline from a rocprofv3 --kernel-trace csv: python tools/timeline.py <kernel_trace.csv> [step_marker_substring].
Prints start offset, duration, queue and short name for every dispatch of the last complete step (delimited by the marker kernel,
default the optimizer kernel), plus the busy / idle split of the main queue."""
import csv
import re
import sys


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return re.sub(r'\(.*$', '', n)[:60]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    marker = sys.argv[2] if len(sys.argv) > 2 else 'adam'
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
    # last two marker groups
    ends = [i for k, i in enumerate(idx) if k + 1 == len(idx) or idx[k + 1] - i > 3]
    a, b = ends[-2] + 1, ends[-1] + 1
    step = rows[a:b]
    t0 = int(step[0]['Start_Timestamp'])
    print('columns', list(rows[0].keys()))
    print(f'step: {len(step)} dispatches, {(int(step[-1]["End_Timestamp"]) - t0) / 1e3:.1f} us')
    for r in step:
        s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
        print(f'{s / 1e3:8.1f} {(e - s) / 1e3:7.1f} q{r.get("Queue_Id", "?")} {short(r["Kernel_Name"])}')


if __name__ == '__main__':
    main()
