"""Static instruction census of one kernel by source line: `hipcc -O3 -gline-tables-only -S --cuda-device-only csrc/uad_gemm.hip -o gemm.s`,
then `python tools/isa_lines.py gemm.s <kernel-name-substring> [top]`.  Attributes every instruction between the kernel's label and its
s_endpgm to the innermost `.loc` line in effect and prints per-line counts by class (MFMA / VALU / SALU / LDS / VMEM / other).  Static
counts equal dynamic counts only for fully unrolled code; the loop structure is visible from the branch targets printed with --blocks."""
import collections
import re
import sys


def klass(m):
    if m.startswith('v_mfma'): return 'mfma'
    if m.startswith('v_'): return 'valu'
    if m.startswith('ds_'): return 'lds'
    if m.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    if m.startswith('s_waitcnt') or m.startswith('s_nop') or m.startswith('s_barrier'): return 'wait'
    if m.startswith('s_'): return 'salu'
    return 'other'


def main():
    path, key = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l) and key in l)
    counts = collections.defaultdict(collections.Counter)
    mnems = collections.defaultdict(collections.Counter)
    cur = ('?', 0)
    files = {int(m.group(1)): m.group(2).split('/')[-1] for m in re.finditer(r'\.file\s+(\d+)\s+(?:"[^"]*"\s+)?"([^"]*)"', '\n'.join(lines))}
    tot = collections.Counter()
    for l in lines[start + 1:]:
        if l.startswith('.Lfunc_end'):
            break
        s = l.strip()
        m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
        if m:
            cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        if not s or s.startswith(('.', ';')) or s.endswith(':'):
            continue
        mn = s.split()[0]
        k = klass(mn)
        counts[cur][k] += 1
        mnems[cur][mn] += 1
        tot[k] += 1
    print('total', dict(tot), 'sum', sum(tot.values()))
    rows = sorted(counts.items(), key=lambda kv: -sum(kv[1].values()))[:top]
    for line, c in rows:
        best = ' '.join(f'{m}:{n}' for m, n in mnems[line].most_common(5))
        print(f'{line[0][-24:]:>24s}:{line[1]:<5d} n={sum(c.values()):5d} ' + ' '.join(f'{k}={c[k]}' for k in ('mfma', 'valu', 'salu', 'lds', 'vmem', 'wait') if c[k]) + '   | ' + best)


if __name__ == '__main__':
    main()
