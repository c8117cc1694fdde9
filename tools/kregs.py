"""Register / LDS / scratch table of the kernels in one translation unit: compiles csrc/<file>.hip to gfx950 assembly
(-gline-tables-only, so tools/isa_lines.py can read the same file) and prints the .amdhsa descriptors of the kernels whose
mangled name contains a key.  `python tools/kregs.py uad_gemm conv5_d16s [more keys]` -> /tmp/isa/uad_gemm.s"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
unit = sys.argv[1]; keys = sys.argv[2:] or ['']
os.makedirs('/tmp/isa', exist_ok=True)
out = f'/tmp/isa/{unit}.s'
src = os.path.join(ROOT, 'unsupervised_anomaly_detection_brain_mri_amd', 'csrc', unit + '.hip')
deps = [src] + [os.path.join(os.path.dirname(src), f) for f in os.listdir(os.path.dirname(src)) if f.endswith(('.inc', '.h'))]
if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value', '-Wno-unused-result',
                           '-gline-tables-only', '-S', '--cuda-device-only', src, '-o', out], stderr=subprocess.DEVNULL)
txt = open(out).read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', txt, re.S):
    name, body = m.group(1), m.group(2)
    if not any(k in name for k in keys): continue
    g = lambda k: (re.search(r'\.amdhsa_' + k + r'\s+(\S+)', body) or [None, '?'])[1]
    print(f"{name[:100]:100s} vgpr {g('next_free_vgpr'):>4s} acc_off {g('accum_offset'):>4s} sgpr {g('next_free_sgpr'):>4s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>4s}")
