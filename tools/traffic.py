"""Post-processes two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; each with --kernel-trace only) of
`bench.py --steps 2 --warmup 1 --no-cpu-baseline` into
  * a per-(kernel, grid) table (stdout / profiles/*.md), and
  * profiles/r0N_traffic_<math>.json, keyed by bench.py's launch-group tags for the big conv kernels
    (tag <-> (kernel template, grid) at the BASELINE configs[1] shapes, N = 64).
FETCH_SIZE / WRITE_SIZE are reported in KiB; FETCH is doubled per MI355X_MICROARCH.md (gfx950 counts 128-B read requests
at 64 B; calibrated here on final_kernel: 134 MB algorithmic read)."""
import collections
import csv
import glob
import json
import sys

fetch_dir, write_dir, out_json = sys.argv[1], sys.argv[2], sys.argv[3]


def load(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            agg[(r['Kernel_Name'], r.get('Grid_Size', ''), r.get('Workgroup_Size', ''))].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


fe, cnt = load(fetch_dir, 'FETCH_SIZE')
wr, _ = load(write_dir, 'WRITE_SIZE')
rows = []
for k in fe:
    rows.append((k, cnt[k], fe[k] * 1024 * 2, wr.get(k, 0.0) * 1024))
rows.sort(key=lambda r: -(r[2] + r[3]) * r[1])
print('| kernel | grid | launches | fetch MB (x2 corrected) | write MB |')
print('|---|---|---|---|---|')
for (name, grid, wg), n, f, w in rows[:40]:
    print(f'| `{name[:80]}` | {grid} | {n} | {f / 1e6:.1f} | {w / 1e6:.1f} |')

# tag -> (kernel substring, grid threads) at N=64, 128x128 (grid = blocks * threads)
MATH = sys.argv[5] if len(sys.argv) > 5 else 'bf16x3'
TAGS = {
    # round 3: the ConvT-class launches run the lane = pixel kernel (uad_conv16s.inc): <TH, TW, CST, WGM, WGN, MF, EPI, PP>
    'dec3.fwd': ('conv5_d16s_kernel<16, 16, 32, 4, 1, 2, 2', 16 * 64 * 256),      # round 6 (late): 16 x 16 tiles, two fragments per wave (UAD_NO_D16S_T16: <8, 16, 32, 4, 1, 1, 2, grid 32 * 64 * 256)
    'dec2.fwd': ('conv5_d16s_kernel<8, 16, 64, 4, 1, 1, 0', 8 * 64 * 256),
    'enc1.dgrad': ('conv5_d16s_kernel<8, 16, 64, 4, 1, 1, 1', 8 * 64 * 256),
    'dec3.dgrad': ('conv5_f16_kernel<8, 16, 16, 4, 1', 32 * 64 * 256),
    'dec2.dgrad': ('conv5_f16_kernel<8, 8, 32, 2, 2', 16 * 64 * 256),
    'enc1.fwd': ('conv5_f16_kernel<8, 8, 32, 2, 2', 16 * 64 * 256),
    'dec3.wgrad': ('conv5_w_bf16_tr_kernel<1, true', 512 * 256),       # round 4: the transpose-read kernel
    'enc1.wgrad': ('conv5_w_bf16_tr_kernel<2, false', 256 * 512),
    'final.fwd+bwd': ('final_kernel<true>', 32 * 64 * 256),
} if MATH == 'bf16x3' else {
    # bf16x6 (round 6): the same three families with three planes per operand; the 64-column F-kind instance walks 16-channel chunks
    'dec3.fwd': ('conv5_d16s_kernel<8, 16, 32, 4, 1, 1, 2', 32 * 64 * 256),
    'dec2.fwd': ('conv5_d16s_kernel<8, 16, 64, 4, 1, 1, 0', 8 * 64 * 256),
    'enc1.dgrad': ('conv5_d16s_kernel<8, 16, 64, 4, 1, 1, 1', 8 * 64 * 256),
    'dec3.dgrad': ('conv5_f16_kernel<8, 16, 16, 4, 1', 32 * 64 * 256),
    'dec2.dgrad': ('conv5_f16_kernel<8, 8, 16, 2, 2', 16 * 64 * 256),
    'enc1.fwd': ('conv5_f16_kernel<8, 8, 16, 2, 2', 16 * 64 * 256),
    'dec3.wgrad': ('conv5_w_bf16_tr_kernel<1, true', 512 * 256),
    'enc1.wgrad': ('conv5_w_bf16_tr_kernel<2, false', 256 * 512),
} if MATH == 'bf16x6' else {
    # exact-fp32 mode (v_mfma_f32_32x32x2_f32 kernels)
    'dec3.fwd': ('conv5_d_kernel<8, 16, 32, 4, 1, true', 32 * 64 * 256),       # round 4: the instance with the fused final epilogue
    'dec3.dgrad': ('conv5_f_kernel<8, 16, 16, 4, 1', 32 * 64 * 256),
    'dec2.dgrad': ('conv5_f_kernel<8, 8, 32, 2, 2', 16 * 64 * 256),
    'enc1.fwd': ('conv5_f_kernel<8, 8, 32, 2, 2', 16 * 64 * 256),
    'dec3.wgrad': ('conv5_w_kernel', 512 * 256),
    'final.fwd+bwd': ('final_kernel<true>', 32 * 64 * 256),      # (only with UAD_NO_FUSED_FINAL_F32=1 since round 4)
}
out = {}
for tag, (sub, grid) in TAGS.items():
    # (the filter-gradient launches' grids follow the planner's slab target: match those by kernel name alone)
    cands = [(k, n, f, w) for (k, n, f, w) in rows if sub in k[0] and (str(grid) == str(k[1]) or tag.endswith('.wgrad'))]
    if cands:
        k, n, f, w = max(cands, key=lambda c: c[2] + c[3])
        out[tag] = {'kernel': sub, 'fetch_bytes': f, 'write_bytes': w, 'launches': n}
out['_commit'] = sys.argv[4] if len(sys.argv) > 4 else 'unknown'
out['_math'] = MATH
out['_note'] = ('HBM bytes per launch from rocprofv3 PMC on MI355X (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes with '
                '--kernel-trace only); FETCH_SIZE doubled per MI355X_MICROARCH.md (calibrated on final_kernel). enc1.fwd and '
                'dec2.dgrad share a kernel template and grid: the entry is the larger of the two.')
json.dump(out, open(out_json, 'w'), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != '_note'}, indent=1))
