"""Per-workload rocprofv3 evidence file for bench.py lines other than the default one (ceVAE, spatial-GMVAE restoration, f-AnoGAN):
    python tools/evidence.py FETCH_DIR WRITE_DIR KERNEL_STATS_CSV OUT.json COMMIT "COMMAND"
FETCH_DIR / WRITE_DIR: output directories of two `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of COMMAND (each with --kernel-trace
only); KERNEL_STATS_CSV: the *kernel_stats.csv of a `rocprofv3 --kernel-trace --stats` run of COMMAND.  OUT.json:
    {"traffic": [{name, grid, launches, fetch_bytes, write_bytes}, ...]   per (kernel, grid), FETCH_SIZE x 2 (gfx950 counts 128-B read requests at 64 B:
                                                                          MI355X_MICROARCH.md; calibrated on final_kernel, tools/traffic.py), KiB -> bytes
     "stats":   [{name, calls, avg_ns, total_ns}, ...]                    per kernel name (every launch of a template instance)
     "_commit", "_command"}
bench.py (evidence_for) picks the dominant launch group's kernel out of both lists."""
import collections
import csv
import glob
import json
import sys

fetch_dir, write_dir, stats_csv, out_json, commit, command = sys.argv[1:7]


def load(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                agg[(r['Kernel_Name'], r.get('Grid_Size', ''))].append(float(r['Counter_Value']))
    return agg


fe, wr = load(fetch_dir, 'FETCH_SIZE'), load(write_dir, 'WRITE_SIZE')
traffic = []
for k, v in fe.items():
    w = wr.get(k, [0.0])
    traffic.append({'name': k[0][:160], 'grid': k[1], 'launches': len(v), 'fetch_bytes': sum(v) / len(v) * 1024 * 2, 'write_bytes': sum(w) / len(w) * 1024})
traffic.sort(key=lambda r: -(r['fetch_bytes'] + r['write_bytes']) * r['launches'])
stats = []
try:
    for r in csv.DictReader(open(stats_csv)):
        stats.append({'name': r['Name'][:160], 'calls': int(r['Calls']), 'avg_ns': float(r['AverageNs']), 'total_ns': float(r['TotalDurationNs'])})
except Exception as e:        # no stats run: traffic only
    print('no kernel stats:', e, file=sys.stderr)
json.dump({'_commit': commit, '_command': command, '_note': 'rocprofv3 on MI355X; PMC passes with --kernel-trace only; FETCH_SIZE doubled (gfx950 correction)',
           'traffic': traffic[:48], 'stats': stats[:48]}, open(out_json, 'w'), indent=1)
for r in traffic[:12]:
    print(f"{r['name'][:90]:90s} grid={r['grid']:>9s} x{r['launches']:<4d} fetch {r['fetch_bytes'] / 1e6:8.1f} MB  write {r['write_bytes'] / 1e6:8.1f} MB")
