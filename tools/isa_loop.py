"""Instruction census of the LOOP BODIES of one kernel (blocks the assembler comments mark `Loop Header` / `in Loop: Header=...`), VALU split by the
classes of tools/isa_classes.py: the per-iteration (= per-tile) instruction mix, without prologue and epilogue.
    python tools/isa_loop.py /tmp/isa/uad_gemm.s <mangled-name substring>"""
import collections
import os
import re
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_classes import vclass


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l) and key in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
    loops = collections.OrderedDict()
    cur = None
    for l in lines[start:end]:
        if l.startswith('.LBB'):
            lab = l.split(':')[0]
            m = re.search(r'in Loop: Header=(BB\w+) Depth=(\d+)', l)
            if 'Loop Header' in l:
                cur = lab[2:]
            elif m:
                cur = m.group(1)
            else:
                cur = None
            continue
        if cur is None:
            continue
        s = l.strip()
        if not s or s.startswith(('.', ';')) or s.endswith(':'):
            continue
        d = loops.setdefault(cur, (collections.Counter(), collections.Counter(), collections.defaultdict(collections.Counter)))
        c, o, top = d
        parts = s.split(None, 1); mn = parts[0]; rest = parts[1] if len(parts) > 1 else ''
        if mn.startswith('v_mfma'): o['mfma'] += 1
        elif mn.startswith('v_'): k = vclass(mn, rest); c[k] += 1; top[k][mn] += 1
        elif mn.startswith('ds_'): o['lds'] += 1
        elif mn.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): o['vmem'] += 1
        elif mn.startswith('s_waitcnt'): o['waitcnt'] += 1
        elif mn.startswith('s_nop'): o['nop'] += 1
        elif 'branch' in mn: o['branch'] += 1
        elif mn.startswith('s_barrier'): o['barrier'] += 1
        elif mn.startswith('s_'): o['salu'] += 1
    for lab, (c, o, top) in loops.items():
        v = sum(c.values())
        print(f'loop {lab}: mfma {o["mfma"]} valu {v} ({v / max(o["mfma"], 1):.1f}/mfma) lds {o["lds"]} vmem {o["vmem"]} salu {o["salu"]} waitcnt {o["waitcnt"]} nop {o["nop"]} branch {o["branch"]} barrier {o["barrier"]}')
        for k in ('fp', 'addr', 'move', 'select', 'cvt', 'xlane'):
            print(f'   {k:7s}{c[k]:5d}  ' + ' '.join(f'{m}:{n}' for m, n in top[k].most_common(7)))


if __name__ == '__main__':
    main()
