"""Builds libuad_hip.so (the C-ABI library declared in include/uad_hip.h) for gfx950 with hipcc.

hipcc cross-compiles without a GPU.  The .so is written IN-TREE next to this file so that it travels with the
repo snapshot to the GPU box; objects are cached under build/ and rebuilt only when a source is newer.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, 'libuad_hip.so')
SOURCES = ['uad_gemm.hip', 'uad_gemm_d16s.hip', 'uad_gemm_d16s3.hip', 'uad_misc.hip', 'uad_gmvae.hip', 'uad_gmd.hip', 'uad_bott.hip', 'uad_eval.hip', 'uad_gan.hip', 'uad_model.hip']
ARCH = 'gfx950'


def _hipcc():
    for c in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found (need ROCm to build libuad_hip.so)')


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def build(force=False, verbose=False):
    hipcc = _hipcc()
    objdir = os.path.join(ROOT, 'build', 'uad_hip')
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, 'uad_kernels.h'), os.path.join(ROOT, 'include', 'uad_hip.h'), os.path.join(CSRC, 'uad_gan_kernels.inc'),
               os.path.join(CSRC, 'uad_gan_create.inc'), os.path.join(CSRC, 'uad_conv16s.inc'), os.path.join(CSRC, 'uad_convk16.inc'),
               os.path.join(CSRC, 'uad_conv5_f32.inc'), os.path.join(CSRC, 'uad_conv5_f32w.inc'), os.path.join(CSRC, 'uad_gemm_common.h'), os.path.join(CSRC, 'uad_gemm_d16s_body.inc')]      # the .inc files are textual parts of uad_gan.hip
    objs = []
    # hidden visibility: the library exports exactly the functions include/uad_hip.h declares (its visibility push), none of the C++ launch layer
    flags = [f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', '-fvisibility-inlines-hidden', '-Wno-unused-value', '-Wno-unused-result']
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace('.hip', '.o'))
        if force or _newer([src] + headers, obj):
            cmd = [hipcc] + flags + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd), file=sys.stderr)
            jobs.append(cmd)
        objs.append(obj)
    # the translation units are independent: compile them side by side (uad_gemm.hip alone takes minutes)
    procs = [subprocess.Popen(cmd) for cmd in jobs]
    failed = [cmd for cmd, pr in zip(jobs, procs) if pr.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    if force or _newer(objs + [os.path.join(CSRC, 'libuad_hip.map')], LIB):
        cmd = [hipcc, f'--offload-arch={ARCH}', '-shared', '-fPIC', '-Wl,--version-script=' + os.path.join(CSRC, 'libuad_hip.map')] + objs + ['-o', LIB]
        if verbose:
            print(' '.join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
