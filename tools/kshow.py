import json, sys
r = json.loads(open(sys.argv[1]).read().replace('NaN', '0'))
k = r['kernels']
tags = sys.argv[2:] or list(k)[:30]
print(r['value'], r['ms_per_step'], ' '.join(f"{t}={k[t]['ms']*1e3:.0f}" for t in tags if t in k))
