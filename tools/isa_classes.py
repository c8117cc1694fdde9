"""Static VALU split by instruction class, per kernel, from the gfx950 assembly of one translation unit (the `.s` that
`tools/kregs.py <unit>` writes to /tmp/isa/<unit>.s; the same code the shipped .so carries — same flags but -gline-tables-only).

    python tools/kregs.py uad_gemm conv5_            # (re)generates /tmp/isa/uad_gemm.s
    python tools/isa_classes.py /tmp/isa/uad_gemm.s conv5_d16s conv5_f16 conv5_w_bf16_tr [--md] [--lines N]

Classes (VALU only; MFMA, LDS, VMEM, SALU are printed beside them):
  fp     v_fma/v_mul/v_add/v_sub/v_max/v_min/v_mac/v_pk_* on f32/f16, v_rcp/v_rsq/v_exp/v_log/v_sqrt …  (without DPP/SDWA modifiers)
  addr   integer / address arithmetic: v_add_u32, v_add3, v_lshl_add, v_mad_u32/u64, v_mul_lo/hi, v_lshlrev/lshrrev/ashrrev, v_and/or/xor/bfe/bfi,
         v_add_co/addc_co, v_sub_u32/subrev, v_lshl_or, v_and_or, v_mul_u32_u24, v_mad_u32_u24, v_lshl_add_u64 …
  move   v_mov_b32/b64, v_accvgpr_read/write, v_readlane/readfirstlane/writelane  (register traffic that computes nothing)
  select v_cmp*/v_cmpx*, v_cndmask
  cvt    v_cvt_*, v_perm_b32, v_pack_*, v_alignbit/alignbyte  (format conversion / packing)
  xlane  anything with a DPP modifier, v_permlane*, ds_bpermute/ds_swizzle are counted under LDS, not here
`--lines N` also prints the N source lines with the most non-fp VALU, by class (needs the .loc lines)."""
import collections
import re
import sys

FP_PREFIX = ('v_fma_f', 'v_fmac_f', 'v_mul_f', 'v_add_f', 'v_sub_f', 'v_subrev_f', 'v_max_f', 'v_min_f', 'v_mac_f', 'v_mad_f', 'v_pk_fma_f', 'v_pk_mul_f', 'v_pk_add_f',
             'v_pk_max_f', 'v_pk_min_f', 'v_rcp_', 'v_rsq_', 'v_exp_', 'v_log_', 'v_sqrt_', 'v_fract_', 'v_floor_', 'v_ceil_', 'v_rndne_', 'v_trunc_', 'v_ldexp_', 'v_frexp_',
             'v_dot2', 'v_med3_f', 'v_max3_f', 'v_min3_f', 'v_div_', 'v_sin_', 'v_cos_', 'v_fmaak_f', 'v_fmamk_f', 'v_madak_f', 'v_madmk_f')
MOVE_PREFIX = ('v_mov_b', 'v_accvgpr_', 'v_readlane', 'v_readfirstlane', 'v_writelane', 'v_swap_b', 'v_pk_mov_b')
SELECT_PREFIX = ('v_cmp', 'v_cndmask')
CVT_PREFIX = ('v_cvt_', 'v_perm_b32', 'v_pack_', 'v_alignbit', 'v_alignbyte', 'v_bfm_')
XLANE_PREFIX = ('v_permlane', 'v_mov_b32_dpp')


def vclass(mn, rest):
    if 'dpp' in mn or 'row_' in rest or 'quad_perm' in rest or 'row_bcast' in rest or 'wave_' in rest and 'dpp' in rest:
        return 'xlane'
    if mn.startswith(XLANE_PREFIX): return 'xlane'
    if mn.startswith(SELECT_PREFIX): return 'select'
    if mn.startswith(MOVE_PREFIX): return 'move'
    if mn.startswith(CVT_PREFIX): return 'cvt'
    if mn.startswith(FP_PREFIX): return 'fp'
    return 'addr'


def kernels(lines, keys):
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\w+):', l)
        if m and any(k in m.group(1) for k in keys):
            yield i, m.group(1)


def demangle(names):
    import subprocess
    try:
        out = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.split('\n')
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    md = '--md' in sys.argv
    nlines = int(sys.argv[sys.argv.index('--lines') + 1]) if '--lines' in sys.argv else 0
    if nlines:
        args = [a for a in args if a != str(nlines)]
    path, keys = args[0], args[1:] or ['']
    lines = open(path).read().split('\n')
    files = {int(m.group(1)): m.group(2).split('/')[-1] for m in re.finditer(r'\.file\s+(\d+)\s+(?:"[^"]*"\s+)?"([^"]*)"', '\n'.join(lines))}
    found = list(kernels(lines, keys))
    dm = demangle([n for _, n in found])
    order = ('fp', 'addr', 'move', 'select', 'cvt', 'xlane')
    if md:
        print('| kernel | MFMA | VALU | VALU/MFMA | fp | addr | move | select | cvt | xlane | LDS | VMEM | SALU |')
        print('|---|---|---|---|---|---|---|---|---|---|---|---|---|')
    for start, name in found:
        c = collections.Counter(); other = collections.Counter(); top = collections.defaultdict(collections.Counter)
        byline = collections.defaultdict(collections.Counter)
        cur = ('?', 0)
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
            parts = s.split(None, 1)
            mn, rest = parts[0], parts[1] if len(parts) > 1 else ''
            if mn.startswith('v_mfma'): other['mfma'] += 1
            elif mn.startswith('v_'):
                k = vclass(mn, rest); c[k] += 1; top[k][mn] += 1
                if k != 'fp': byline[cur][k] += 1
            elif mn.startswith('ds_'): other['lds'] += 1
            elif mn.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): other['vmem'] += 1
            elif mn.startswith('s_') and not mn.startswith(('s_waitcnt', 's_nop', 's_barrier')): other['salu'] += 1
        valu = sum(c.values())
        short = re.sub(r'^void ', '', dm.get(name, name)).replace('(anonymous namespace)::', '').split('(')[0]
        pct = lambda k: f'{c[k]} ({100.0 * c[k] / max(valu, 1):.0f} %)'
        if md:
            print(f'| `{short}` | {other["mfma"]} | {valu} | {valu / max(other["mfma"], 1):.1f} | ' + ' | '.join(pct(k) for k in order) + f' | {other["lds"]} | {other["vmem"]} | {other["salu"]} |')
        else:
            print(f'{short}\n  mfma {other["mfma"]}  valu {valu}  ({valu / max(other["mfma"], 1):.1f} per MFMA)  lds {other["lds"]}  vmem {other["vmem"]}  salu {other["salu"]}')
            for k in order:
                print(f'  {k:7s}{pct(k):>14s}   ' + ' '.join(f'{m}:{n}' for m, n in top[k].most_common(6)))
            if nlines:
                for line, cc in sorted(byline.items(), key=lambda kv: -sum(kv[1].values()))[:nlines]:
                    print(f'    {line[0][-22:]:>22s}:{line[1]:<5d} ' + ' '.join(f'{k}={cc[k]}' for k in order if cc[k]))


if __name__ == '__main__':
    main()
