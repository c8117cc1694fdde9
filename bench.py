#!/usr/bin/env python
"""bench.py — VAE train step (fwd + bwd + TF-Adam) on synthetic Brainweb-like slices, one process per GPU.

    python bench.py --gpus 1 --steps 30 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: VAE (variational_autoencoder) 128x128 "bf16", batch 64 per GPU.  Plain bf16
arithmetic cannot meet the north_star parity bar (1e-4 rel vs fp32), so the default math mode is bf16x3: every fp32
operand is split hi+lo into two bf16 values and a product is hi*hi + hi*lo + lo*hi on the bf16 matrix cores with fp32
accumulation (~2^-17 relative error per product; tests/test_gpu_model.py holds it to the same 1e-4 bar as --math f32,
the exact-fp32-MFMA mode, whose number is reported alongside on 1 GPU).  The slice batch x is resident in HBM before the
timed region; eps and the three dropout masks are drawn FRESH every timed step on the device (counter-based generator, one launch, inside
the timed region).  Also reported: `trainer_loop_slices_per_s` = one process() epoch of trainers.VAE (the reference's unit of work), the
roofline of the dominant kernel against BOTH bounds (mfma_fraction, hbm_fraction), the CPU baseline (torch-CPU oneDNN port, all physical
cores).  N>1: slice-batch data parallel, weak scaling (64 slices per GPU), the three
gradient segments are all-reduced over RCCL as soon as each is complete, overlapped with the rest of the backward.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process launches, before the HIP runtime initialises (see the package's __init__): the engine's two streams and RCCL's need their own hardware queues
if int(os.environ.get('WORLD_SIZE', '1') or 1) > 1 or os.environ.get('UAD_BENCH_REHEARSAL'):
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

H = W = 128
BATCH = 64
ZDIM = 128
INTER = 8
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: bf16 dense MFMA peak
PEAK_HBM_GBS = 8000.0


_REAL_STDOUT = None


def claim_stdout():
    """Rank 0 prints ONE JSON line on stdout -- and nothing else may: RCCL writes a start-up banner (version, host, library path) to file descriptor 1
    when the first communicator of a process is created (torch.distributed's and the library's own alike).  From here on fd 1 points at stderr for
    native code and Python alike; emit_json() writes the line to the real stdout."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit_json(obj):
    line = (json.dumps(obj) + '\n').encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(line.decode()); sys.stdout.flush()
    else:
        sys.stdout.flush()
        os.write(_REAL_STDOUT, line)


def conv_layers():
    """(tag-prefix, positions_per_slice, taps*CB*CS) of every k5 s2 conv block; MACs/slice = positions * taps*CB*CS."""
    layers = []
    cin, res = 1, H
    i = 0
    while res > INTER:
        f = min(128, 32 * 2 ** i)
        layers.append((f'enc{i}', (res // 2) ** 2, 25 * cin * f))
        cin, res, i = f, res // 2, i + 1
    i = 0
    while res < H:
        f = max(32, 128 // 2 ** i)
        layers.append((f'dec{i}', res * res, 25 * f * cin))
        cin, res, i = f, res * 2, i + 1
    return layers


def flops_per_tag(n):
    fl = {}
    for name, pos, k in conv_layers():
        f = 2.0 * n * pos * k
        for kind in ('fwd', 'dgrad', 'wgrad'):
            fl[f'{name}.{kind}'] = f
    return fl


def bytes_per_tag(n, fin_bits=True):
    """Algorithmic HBM bytes of one launch of every conv launch group: fp32 activations read / written once per
    kernel; weights (<= 1.6 MB) ignored.  fwd: input + output; dgrad: d_out + the producer's pre-BN output (activation backward in the
    epilogue) + d_in; wgrad: layer input + d_out.  The last decoder block keeps its d loss / d c as one pattern word + one float per
    output pixel (fin_bits): its forward writes x_hat, L1 and those 8 bytes per pixel instead of 32 floats."""
    by = {}
    cin, res = 1, H
    layers = []
    i = 0
    while res > INTER:
        f = min(128, 32 * 2 ** i)
        layers.append((f'enc{i}', res * res * cin, (res // 2) ** 2 * f))       # floats per slice: layer input, layer output
        cin, res, i = f, res // 2, i + 1
    i = 0
    while res < H:
        f = max(32, 128 // 2 ** i)
        layers.append((f'dec{i}', res * res * cin, (2 * res) ** 2 * f))
        cin, res, i = f, res * 2, i + 1
    last = layers[-1][0]
    for name, fin, fout in layers:
        b_in, b_out = 4.0 * n * fin, 4.0 * n * fout
        d_out = b_out
        if name == last and fin_bits:
            d_out = 8.0 * n * H * W
            by[f'{name}.fwd'] = b_in + d_out + 12.0 * n * H * W        # + target read, x_hat and L1 written
        elif name == last:
            # exact-fp32 mode (round 4: final conv + loss fused there too): d loss / d c leaves as fp32, in place of the block's output
            by[f'{name}.fwd'] = b_in + b_out + 12.0 * n * H * W
        else:
            by[f'{name}.fwd'] = b_in + b_out
        by[f'{name}.dgrad'] = d_out + 2.0 * b_in
        by[f'{name}.wgrad'] = b_in + d_out
    return by


def spec_conv_tags(spec, h, n):
    """{launch-group tag: (flop, algorithmic bytes)} of the k x k stride-2 conv / transposed-conv layers of ANY graph of the family, from the handle's
    tensor table (names `...enc_conv2D_<i>/kernel` [k,k,cin,cout], `...dec_Conv2DT_<i>/kernel` [k,k,cout,cin]; models/customlayers.py:16-38): the
    spatial GMVAE's bench line takes its roofline object from this (flops_per_tag / bytes_per_tag above are the VAE's closed forms; a CPU test holds the
    two against each other).  fwd: input + output; dgrad: d_out + the producer's pre-BN output + d_in; wgrad: layer input + d_out -- fp32, read / written once."""
    import re
    enc, dec = {}, {}
    for name, shape, *_ in spec:
        m = re.search(r'enc_conv2D_(\d+)/kernel$', name)
        if m:
            enc[int(m.group(1))] = tuple(shape)
        m = re.search(r'dec_Conv2DT_(\d+)/kernel$', name)
        if m:
            dec[int(m.group(1))] = tuple(shape)
    out = {}
    res = h
    for i in sorted(enc):
        k, _, cin, cout = enc[i]
        big, small = res * res, (res // 2) ** 2
        f = 2.0 * n * small * k * k * cin * cout
        b_in, b_out = 4.0 * n * big * cin, 4.0 * n * small * cout
        out[f'enc{i}.fwd'] = (f, b_in + b_out)
        out[f'enc{i}.dgrad'] = (f, b_out + 2.0 * b_in)
        out[f'enc{i}.wgrad'] = (f, b_in + b_out)
        res //= 2
    for i in sorted(dec):
        k, _, cout, cin = dec[i]
        small, big = res * res, (2 * res) ** 2
        f = 2.0 * n * small * k * k * cin * cout
        b_in, b_out = 4.0 * n * small * cin, 4.0 * n * big * cout
        out[f'dec{i}.fwd'] = (f, b_in + b_out)
        out[f'dec{i}.dgrad'] = (f, b_out + 2.0 * b_in)
        out[f'dec{i}.wgrad'] = (f, b_in + b_out)
        res *= 2
    return out


def last_block_kernel(tag, nblocks, math='bf16x3'):
    """Kernel-template substring of the three launch groups of the LAST decoder block (the dominant kernels of every AE-family graph) in the committed
    rocprofv3 files; None for any other tag (several layers share a template there)."""
    if tag.split('.')[0] != f'dec{nblocks - 1}':
        return None
    kind = tag.split('.')[1]
    if math == 'f32':
        return {'fwd': 'conv5_d_kernel<8, 16, 32, 4, 1, true', 'dgrad': 'conv5_f_kernel<8, 16, 16, 4, 1', 'wgrad': 'conv5_w_kernel'}.get(kind)
    # (round 6: the bf16x3 training instance of the fused-final ConvT kernel runs 16 x 16 tiles, two fragments per wave; bf16x6 keeps 8 x 16)
    fwd = 'conv5_d16s_kernel<16, 16, 32, 4, 1, 2, 2' if math == 'bf16x3' else 'conv5_d16s_kernel<8, 16, 32, 4, 1, 1, 2'
    return {'fwd': fwd, 'dgrad': 'conv5_f16_kernel<8, 16, 16, 4, 1', 'wgrad': 'conv5_w_bf16_tr_kernel<1, true'}.get(kind)


def evidence_for(workload, kernel_substr, flop, peak):
    """(traffic, rocprof) objects of a roofline line from profiles/r06_evidence_<workload>.json (r05_ as fallback) (tools/evidence.py: PMC FETCH_SIZE / WRITE_SIZE passes and the
    rocprofv3 --kernel-trace --stats summary of the SAME bench command), for the kernel whose name contains kernel_substr; (None, None) when absent."""
    if not kernel_substr:
        return None, None
    doc = fname = None
    for rnd in ('r06', 'r05'):
        try:
            fname = f'{rnd}_evidence_{workload}.json'
            doc = json.load(open(os.path.join(ROOT, 'profiles', fname)))
            break
        except Exception:
            doc = None
    if doc is None:
        return None, None
    src = (f"profiles/{fname}: `{doc.get('_command')}` under rocprofv3 at commit {doc.get('_commit')}; not re-measured in this run "
           "(PMC counters need the profiler)")
    traffic = rocprof = None
    rows = [r for r in doc.get('traffic', []) if kernel_substr in r['name']]
    if rows:
        r = max(rows, key=lambda r: r['fetch_bytes'] + r['write_bytes'])
        traffic = {'bytes': int(r['fetch_bytes'] + r['write_bytes']), 'fetch_bytes': int(r['fetch_bytes']), 'write_bytes': int(r['write_bytes']),
                   'launches_averaged': r['launches'], 'source': src + '; --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH x2 (gfx950 correction)'}
    rows = [r for r in doc.get('stats', []) if kernel_substr in r['name']]
    if rows:
        r = max(rows, key=lambda r: r['total_ns'])
        avg_ms = r['avg_ns'] * 1e-6
        rocprof = {'avg_launch_ms': round(avg_ms, 4), 'calls': r['calls'], 'frac': round(flop / (avg_ms * 1e-3) / 1e12 / peak, 4), 'source': src + '; --kernel-trace --stats'}
    return traffic, rocprof


def train_flops_per_slice():
    """371.5 M MAC fwd -> 0.743 GFLOP; train step = 3x fwd minus the enc0 data-grad."""
    fwd_macs = sum(pos * k for _, pos, k in conv_layers())
    small = 128 * 16 * 64 + 2 * 1024 * 128 + 128 * 1024 + 16 * 128 * 64 + 32 * H * W
    return 2.0 * (3 * (fwd_macs + small) - conv_layers()[0][1] * conv_layers()[0][2])


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _physical_cores():
    try:
        import psutil
        return int(psutil.cpu_count(logical=False) or os.cpu_count() or 1)
    except Exception:
        return int(os.cpu_count() or 1)


def cpu_baseline(batch=64, steps=20, warmup=5):
    """SURVEY.md section 8d's CPU baseline: the reference's TF-CPU path cannot be run (tensorflow==1.15.2 is not installable here and reference
    Python may not travel to the GPU box), so the SAME step -- VAE forward + losses + backward + TF-Adam, fp32, explicit TF-SAME padding,
    dropout masks and eps as inputs -- is timed as a torch-CPU graph (oneDNN convolutions, autograd; the formulation of tests/torch_ref.py,
    which the tests hold against the oracle) on all physical cores of this host: median of `steps` steps after `warmup`, batch 64.
    kind = "port"; cores and the CPU model are stated."""
    import torch
    from oracle import nn as onn, vae as ovae
    from tests import torch_ref as tr
    cores = _physical_cores()
    prev = torch.get_num_threads()
    torch.set_num_threads(cores)
    m = ovae.Model('VAE', H, W, 1, INTER, ZDIM)
    p_np = ovae.init_params(m.spec, seed=3)
    params = tr.to_torch(p_np, dtype=torch.float32, requires_grad=True)
    names = [n for n, _, _ in m.spec]
    slots_m = {n: torch.zeros_like(params[n]) for n in names}
    slots_v = {n: torch.zeros_like(params[n]) for n in names}
    x = torch.from_numpy(ovae.synthetic_slices(batch, H, W, seed=0))
    rng = np.random.default_rng(1)
    lr, b1, b2, eps_adam = 1e-4, 0.5, 0.999, 1e-8
    n_pool = int(np.log2(H) - np.log2(INTER))
    times = []
    for t in range(1, warmup + steps + 1):
        eps = torch.from_numpy(rng.standard_normal((batch, ZDIM)).astype(np.float32))
        masks = {k: torch.from_numpy(onn.make_dropout_mask(rng, shp, 0.2)) for k, shp in
                 (('mu', (batch, ZDIM)), ('sigma', (batch, ZDIM)), ('dec', (batch, INTER * INTER * 16)))}
        t0 = time.perf_counter()
        losses, _, _ = tr.forward_loss('VAE', m.spec, params, x, eps, masks, INTER, n_pool)
        grads = torch.autograd.grad(losses['loss'], [params[n] for n in names])
        with torch.no_grad():
            lr_t = lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
            for n, g in zip(names, grads):
                slots_m[n].mul_(b1).add_(g, alpha=1.0 - b1)
                slots_v[n].mul_(b2).addcmul_(g, g, value=1.0 - b2)
                params[n].sub_(lr_t * slots_m[n] / (slots_v[n].sqrt() + eps_adam))
        dt = time.perf_counter() - t0
        if t > warmup:
            times.append(dt)
    torch.set_num_threads(prev)
    med = float(np.median(times))
    return {'value': round(batch / med, 2), 'unit': 'slices/s', 'cores': cores, 'kind': 'port', 'cpu_model': _cpu_model(),
            'ms_per_step': round(med * 1e3, 2),
            'sample': f'median of {steps} fp32 VAE train steps (after {warmup} warm-up) at batch {batch} as a torch-CPU oneDNN graph with '
                      f'explicit TF-SAME padding (tests/torch_ref.py), {cores} threads = all physical cores of {os.cpu_count()} logical CPUs; '
                      f'TF-CPU 1.15 itself is not installable here; {sum(times):.1f} s of CPU work'}


def bench_gmvae(args):
    """BASELINE.json configs[4]: spatial GMVAE 256x256 restoration-mode inference, slices sharded over the GPUs (no
    collective on the data path: replicas with different slices).  One 'step' = restore_steps restoration iterations
    (forward + data-only backward of loss + tv*TV + in-place update, all on device) for a batch of slices; value =
    restored slices per second."""
    import torch
    import torch.distributed as dist
    from unsupervised_anomaly_detection_brain_mri_amd.engine import Engine
    from unsupervised_anomaly_detection_brain_mri_amd.utils.synthetic import synthetic_slices
    world = int(os.environ.get('WORLD_SIZE', '1')); rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    rehearsal = bool(os.environ.get('UAD_BENCH_REHEARSAL'))
    if os.environ.get('UAD_BENCH_REHEARSAL'):
        # several processes on ONE GPU: the fused bottleneck's groups of four sibling workgroups (uad_bott.hip) need all four resident at once; two processes'
        # kernels can hold each other's slots until the bounded exchange gives up (it reports, it does not hang).  The one-workgroup-per-sample form has no
        # inter-workgroup wait.  (One process per GPU -- the deployment this library is written for -- is unaffected.)
        os.environ.setdefault('UAD_BOTT_Q1', '1')    # single-GPU rehearsal of the multi-process path: every rank on cuda:0, gloo instead of RCCL
    if rehearsal:
        local_rank = 0
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo', rank=rank, world_size=world) if rehearsal else dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
    hh, bs, rs = 256, BATCH, args.restore_steps
    eng = Engine('GMVAE_spatial', hh, hh, 1, 8, max_batch=bs, device=f'cuda:{local_rank}', math=args.math, dim_c=9, dim_z=1, dim_w=1)
    rng = np.random.default_rng(3)
    flat = np.zeros(eng.nparams, np.float32)
    for name, shape, off in eng.spec:
        cnt = int(np.prod(shape))
        if name.endswith('kernel'):
            rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
            lim = np.sqrt(6.0 / (shape[-2] * rf + shape[-1] * rf))
            flat[off:off + cnt] = rng.uniform(-lim, lim, cnt)
        elif name.endswith('gamma'):
            flat[off:off + cnt] = 1.0
        elif name == 'Variable':
            flat[off:off + cnt] = 0.1
    eng.set_params(flat)
    x0 = torch.from_numpy(synthetic_slices(bs, hh, hh, seed=1000 + rank)).cuda()
    g = torch.Generator(device='cuda').manual_seed(1 + rank)
    e_w = torch.randn(bs, 8, 8, 1, device='cuda', generator=g); e_z = torch.randn(bs, 8, 8, 1, device='cuda', generator=g)

    def step():
        xr = x0.clone()
        for _ in range(rs):
            eng.restore_step(xr, e_w, e_z, tv_lambda=1.8, restore_lr=1e-3)
        return xr

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        xr = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(xr).all()
    eng.profile(True)
    xr = x0.clone()
    for _ in range(5):
        eng.restore_step(xr, e_w, e_z, tv_lambda=1.8, restore_lr=1e-3)
    rep = eng.profile_report()
    eng.profile(False)
    if rank == 0:
        value = bs * world * args.steps / dt
        flop_slice_step = 4.88e9          # fwd + data-only bwd at 256x256
        res = {'metric': f'MRI slices/sec GMVAE-spatial restoration ({rs} steps, 256x256, bs={bs}/GPU)', 'value': round(value, 2),
               'unit': 'slices/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
               'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
               'dtype': args.math, 'data': 'synthetic',
               'config': {'workload': f'BASELINE.json configs[4]: spatial GMVAE 256x256x1, {rs} restoration steps per slice '
                                      f'(forward + data-gradient backward of loss + 1.8*TV + in-place update, on device), '
                                      f'{bs} slices per GPU per step, dim_c 9 dim_z 1 dim_w 1',
                          'ms_per_restore_iteration': round(dt / args.steps / rs * 1e3, 4),
                          'algorithmic_tflops': round(value * rs * flop_slice_step / 1e12, 2), 'parallelism': f'replicas{world}'},
               'kernels': {t: {'ms': round(ms / c, 4)} for t, (c, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1])}}
        # roofline of the restoration iteration's dominant conv launch group, measured live (HIP events around the group, profiling pass above)
        tags = spec_conv_tags(eng.spec, hh, bs)
        conv = {t: ms / c for t, (c, ms) in rep.items() if t in tags}
        if conv:
            dom = max(conv, key=conv.get)
            fl, by = tags[dom]
            peak = PEAK_F32_MFMA_TFLOPS if args.math == 'f32' else PEAK_BF16_MFMA_TFLOPS / 3.0
            alg, gbs = fl / (conv[dom] * 1e-3) / 1e12, by / (conv[dom] * 1e-3) / 1e9
            nblk = len([1 for name, *_ in eng.spec if 'dec_Conv2DT_' in name and name.endswith('/kernel')])
            ev_traffic, ev_rocprof = evidence_for(f'gmvae_restore_b{bs}', last_block_kernel(dom, nblk, args.math), fl, peak) if args.math != 'f32' else (None, None)
            for t in res['kernels']:
                if t in tags:
                    res['kernels'][t]['tflops'] = round(tags[t][0] / (conv[t] * 1e-3) / 1e12, 2)
            res['roofline'] = {'bound': 'mfma', 'kernel': dom, 'achieved': round(alg, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': round(alg / peak, 4),
                               'mfma_fraction': round(alg / peak, 4), 'hbm_fraction': round(gbs / PEAK_HBM_GBS, 4), 'hbm_achieved_gbs': round(gbs, 1),
                               'hbm_peak_gbs': PEAK_HBM_GBS, 'algorithmic_bytes_per_launch': int(by), 'algorithmic_flop_per_launch': int(fl),
                               'traffic': ev_traffic, 'rocprof': ev_rocprof, 'avg_launch_ms': round(conv[dom], 4),
                               'instruction': 'v_mfma_f32_32x32x2_f32 (exact fp32)' if args.math == 'f32' else
                                              '3 x v_mfma_f32_32x32x16_bf16 per fp32 product; peak = dense bf16 MFMA 2500 TFLOP/s / 3 products',
                               'note': 'restoration iteration = forward + data-gradient backward: no filter gradients'}
        emit_json(res)
    if world > 1:
        dist.destroy_process_group()


def gmvae_volume_cpu_baseline(hh, rs, slices):
    """CPU leg of the full-volume line: the oracle's restoration gradient (numpy fp64 restatement of trainers/GMVAE_spatial.py:91-92, 168-199) on ONE
    256 x 256 slice for a bounded number of iterations, plus the host post-processing the reference runs per volume (scipy erosion of every brain
    mask, 5 x 5 x 5 median filter: utils/Evaluation.py:84-127) on the full volume size -- scaled to slices/s of the whole pipeline.  kind = "port"."""
    import scipy.ndimage
    from oracle import gmvae as og
    m = og.GMVAE(hh, hh, 1, 8, 9, 1, 1, 1.0)
    p = {k: v.astype(np.float64) for k, v in og.init_params(m.spec, seed=3).items()}
    rng = np.random.default_rng(0)
    x = rng.random((1, hh, hh, 1))
    e_w, e_z = rng.standard_normal((1, 8, 8, 1)), rng.standard_normal((1, 8, 8, 1))
    m.restore_grads(p, x, e_w, e_z, 1.8)                      # warm-up
    it, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 12.0 and it < 50:
        x = x - 1e-3 * m.restore_grads(p, x, e_w, e_z, 1.8)
        it += 1
    t_iter = (time.perf_counter() - t0) / max(it, 1)
    vol = rng.random((slices, hh, hh)) * (rng.random((slices, hh, hh)) < 0.4)
    yy, xx = np.mgrid[0:hh, 0:hh]
    mask = ((yy - hh / 2) / (0.4 * hh)) ** 2 + ((xx - hh / 2) / (0.36 * hh)) ** 2 <= 1
    t1 = time.perf_counter()
    strel = scipy.ndimage.generate_binary_structure(2, 1)
    for _ in range(slices):
        scipy.ndimage.binary_erosion(mask, structure=strel, iterations=12)
    scipy.ndimage.median_filter(vol, (5, 5, 5))
    t_post = time.perf_counter() - t1
    per_volume = slices * rs * t_iter + t_post
    return {'value': round(slices / per_volume, 5), 'unit': 'slices/s', 'cores': 1, 'kind': 'port', 'cpu_model': _cpu_model(),
            'seconds_per_volume': round(per_volume, 1),
            'sample': f'{it} restoration iterations of ONE {hh} x {hh} slice with the numpy fp64 oracle ({t_iter:.3f} s each; a volume needs {slices} x {rs}) + scipy '
                      f'erosion of {slices} masks and the 5x5x5 median filter of the [{slices},{hh},{hh}] volume ({t_post:.1f} s), one thread; '
                      f'{time.perf_counter() - t0:.1f} s of CPU work'}


def bench_gmvae_volume(args):
    """BASELINE.json configs[4] as the config states it -- FULL-VOLUME restoration-mode inference: one synthetic patient per GPU ([110, 256, 256]) through
    the product path of utils/Evaluation.py:223-312 + trainers/GMVAE_spatial.py:168-199 -- GMVAE_spatial.reconstruct() (restore_steps restoration iterations,
    slices batched 16 at a time, on device) -> uad_residual (brain-masked positive residual, hyper-intensity prior) -> uad_erode_cross -> uad_median3d ->
    uad_scores_* (AUROC / AUPRC / Dice sweep) -> uad_cc_filter.  Patients are sharded over the ranks (Evaluation._sharded_map: no collective on the
    restoration path, the residual volumes are exchanged for the pooled metrics).  One 'step' = one volume per GPU; value = slices per second."""
    import tempfile
    import torch
    import torch.distributed as dist
    from unsupervised_anomaly_detection_brain_mri_amd.models import gaussian_mixture_variational_autoencoder_spatial as net
    from unsupervised_anomaly_detection_brain_mri_amd.trainers import GMVAE_spatial
    from unsupervised_anomaly_detection_brain_mri_amd.utils import Evaluation
    from unsupervised_anomaly_detection_brain_mri_amd.utils.default_config_setup import get_options, get_config
    from unsupervised_anomaly_detection_brain_mri_amd.utils.synthetic import SyntheticDataset, synthetic_slices
    world = int(os.environ.get('WORLD_SIZE', '1')); rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    rehearsal = bool(os.environ.get('UAD_BENCH_REHEARSAL'))
    if rehearsal:
        os.environ.setdefault('UAD_BOTT_Q1', '1')
        local_rank = 0
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo', rank=rank, world_size=world) if rehearsal else dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
    hh, bs, rs, S = 256, BATCH, args.restore_steps, args.volume_slices
    tmp = tempfile.mkdtemp(prefix='uad_bench_')
    opt = get_options(batchsize=bs, learningrate=5e-5, numEpochs=1, zDim=128, outputWidth=hh, outputHeight=hh,
                      config={'CHECKPOINTDIR': os.path.join(tmp, 'ck'), 'SAMPLEDIR': os.path.join(tmp, 'smp')})
    cfg = get_config(GMVAE_spatial, opt, 'ADAM', [8, 8], 0.2, SyntheticDataset(4, 4, hh, hh, seed=0))
    cfg.dim_c, cfg.dim_z, cfg.dim_w, cfg.restore_steps, cfg.restore_lr, cfg.tv_lambda = 9, 1, 1, rs, 1e-3, 1.8
    model = GMVAE_spatial(None, cfg, network=net, world=1, device=f'cuda:{local_rank}')      # (inference only: no gradient all-reduce to set up)
    eng = model.engine
    eng.set_math(args.math)
    # one synthetic patient per rank: phantom slices with planted hyper-intense lesions, their label map and the brain mask
    volumes, labels, masks = [], [], []
    for k in range(world):
        img, lab, msk = synthetic_slices(S, hh, hh, seed=2000 + k, lesions=True)
        volumes.append(img[..., 0].astype(np.float32)); labels.append(lab.astype(np.uint8)); masks.append(msk.astype(np.float32))
    options = dict(opt)
    options.update({'keepOnlyPositiveResiduals': True, 'applyHyperIntensityPrior': True, 'medianFiltering': True, 'erodeBrainmask': True, 'threshold': 'bestdice'})

    def step():
        return Evaluation.evaluate_arrays(volumes, labels, masks, model, options)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ev = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # phase shares (untimed pass, own patient): restoration | residual + erosion + median | scoring (sort, scans, Dice sweep, small-component filter)
    ph = {}
    k = rank
    torch.cuda.synchronize(); t1 = time.perf_counter()
    rec = np.concatenate([model.reconstruct(volumes[k][s0:s0 + bs, ..., None])['reconstruction'] for s0 in range(0, S, bs)])
    torch.cuda.synchronize(); ph['restoration_s'] = time.perf_counter() - t1
    t1 = time.perf_counter()
    em = eng.erode_cross(masks[k], 12)
    d, _ = eng.residual(volumes[k][..., None], rec, em[..., None], pos_only=True, prior_thresh=float(np.quantile(volumes[k], 0.9)))
    dm = eng.median3d(d[..., 0], 5)
    torch.cuda.synchronize(); ph['residual_erode_median_s'] = time.perf_counter() - t1
    t1 = time.perf_counter()
    Evaluation._score_diffs(model, [dm], [labels[k]], options)
    torch.cuda.synchronize(); ph['scoring_s'] = time.perf_counter() - t1
    eng.profile(True)
    xr = torch.from_numpy(volumes[k][:bs, ..., None].copy()).to(eng.device)
    g = torch.Generator(device='cuda').manual_seed(1 + rank)
    e_w = torch.randn(bs, 8, 8, 1, device='cuda', generator=g); e_z = torch.randn(bs, 8, 8, 1, device='cuda', generator=g)
    for _ in range(5):
        eng.restore_step(xr, e_w, e_z, tv_lambda=1.8, restore_lr=1e-3)
    rep = eng.profile_report()
    eng.profile(False)
    if rank == 0:
        per_volume = dt / args.steps
        value = S * world * args.steps / dt
        tot = sum(ph.values())
        res = {'metric': f'MRI slices/sec GMVAE-spatial full-volume restoration inference ({rs} steps, {S} x 256x256 per volume, {bs} slices in flight)',
               'value': round(value, 2), 'unit': 'slices/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
               'ms_per_step': round(per_volume * 1e3, 2), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.math, 'data': 'synthetic',
               'config': {'workload': f'BASELINE.json configs[4]: spatial GMVAE 256x256x1 full-volume restoration-mode inference -- one [{S},256,256] patient per GPU: '
                                      f'reconstruct() = {rs} restoration iterations per slice ({bs} slices in flight) -> masked positive residual + hyper-intensity prior '
                                      f'-> 12 x cross erosion of the brain masks -> 5x5x5 median -> AUROC / AUPRC / greedy Dice sweep -> <= 7-voxel component filter '
                                      f'(utils/Evaluation.py:223-312, trainers/GMVAE_spatial.py:168-199), dim_c 9 dim_z 1 dim_w 1',
                          'seconds_per_volume': round(per_volume, 3), 'parallelism': f'patients{world}',
                          'phase_seconds_one_volume': {k2: round(v, 4) for k2, v in ph.items()},
                          'post_processing_share': round((tot - ph['restoration_s']) / tot, 4),
                          'metrics_of_the_random_init_model': {'diff_AUC': round(float(ev['diff_AUC']), 6), 'diff_AUPRC': round(float(ev['diff_AUPRC']), 6),
                                                               'bestDiceScore': round(float(ev['bestDiceScore']), 6)}},
               'kernels': {t: {'ms': round(ms / c, 4)} for t, (c, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1])}}
        tags = spec_conv_tags(eng.spec, hh, bs)
        conv = {t: ms / c for t, (c, ms) in rep.items() if t in tags}
        if conv:
            dom = max(conv, key=conv.get)
            fl, by = tags[dom]
            peak = PEAK_F32_MFMA_TFLOPS if args.math == 'f32' else PEAK_BF16_MFMA_TFLOPS / 3.0
            alg, gbs = fl / (conv[dom] * 1e-3) / 1e12, by / (conv[dom] * 1e-3) / 1e9
            nblk = len([1 for name, *_ in eng.spec if 'dec_Conv2DT_' in name and name.endswith('/kernel')])
            ev_traffic, ev_rocprof = evidence_for(f'gmvae_restore_b{bs}', last_block_kernel(dom, nblk, args.math), fl, peak) if args.math != 'f32' else (None, None)
            res['roofline'] = {'bound': 'mfma', 'kernel': dom, 'achieved': round(alg, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': round(alg / peak, 4),
                               'hbm_fraction': round(gbs / PEAK_HBM_GBS, 4), 'algorithmic_bytes_per_launch': int(by), 'algorithmic_flop_per_launch': int(fl),
                               'traffic': ev_traffic, 'rocprof': ev_rocprof, 'avg_launch_ms': round(conv[dom], 4),
                               'note': 'dominant launch of the restoration iteration (forward + data-gradient backward), which is '
                                       f'{100 * ph["restoration_s"] / tot:.1f} % of a volume; same kernel as the restoration-only line'}
        if not args.no_cpu_baseline and not args.quick:
            res['cpu_baseline'] = gmvae_volume_cpu_baseline(hh, rs, S)
        emit_json(res)
    if world > 1:
        dist.destroy_process_group()


def gan_macs(variant, h, zdim=128, dim=64):
    """Multiply-accumulates per sample and forward pass of the encoder, generator and critic."""
    if variant == 'resnet':
        r = h // 8
        enc = sum((h >> (i + 1)) ** 2 * 25 * ci * co for i, (ci, co) in enumerate(((1, 32), (32, 64), (64, 128)))) + r * r * 128 * zdim
        gen, c, res = zdim * r * r * 8 * dim, 8 * dim, r
        for f, st in ((8 * dim, 1), (4 * dim, 2), (2 * dim, 2), (dim, 2)):
            gen += res * res * 9 * c * f                      # conv1 at the input resolution
            gen += res * res * 9 * f * f                      # transposed conv2: 9 taps per INPUT position
            if st == 2:
                gen += res * res * c * f                      # k1 s2 shortcut
            c, res = f, res * st
        gen += res * res * c
        dis, c, res = h * h * 9 * dim, dim, h
        for f, st in ((2 * dim, 2), (4 * dim, 2), (8 * dim, 2), (8 * dim, 1)):
            dis += res * res * 9 * c * f + (res // st) ** 2 * 9 * f * f
            if st == 2:
                dis += res * res * c * f
            c, res = f, res // st
        return enc, gen, dis + res * res * c
    npool = int(np.log2(h)) - 3
    chans = [min(128, 32 * 2 ** i) for i in range(npool)]
    enc = dis = 0
    c, res = 1, h
    for f in chans:
        res //= 2
        enc += res * res * 25 * c * f
        c = f
    dis = enc + 64 * c
    enc += 64 * c * (c // 8) + 64 * (c // 8) * zdim
    gen = zdim * 64 * (c // 8) + 64 * (c // 8) * c
    res = 8
    for i in range(npool):
        f = max(32, 128 >> i)
        gen += res * res * 25 * c * f
        c, res = f, res * 2
    return enc, gen + res * res * c, dis


def gan_cpu_baseline(variant, hh, zd):
    """The numpy oracle (kind "port") on a bounded sample: ONE WGAN-GP batch iteration (1 generator + 5 critic phases, gradients
    only) at batch 1, fp32, BLAS threads capped like cpu_baseline()."""
    threads = max(1, min(int(os.environ.get('UAD_CPU_THREADS', '32')), os.cpu_count() or 1))
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=threads)
    except Exception:
        limiter = None
    from oracle import vae as ovae
    if variant == 'resnet':
        from oracle import fanogan_schlegl as ofs
        m = ofs.FAnoGANSchlegl(hh, hh // 8, zd, 64)
    else:
        from oracle import fanogan as ofa
        m = ofa.FAnoGAN(hh, 8, zd)
    p = ovae.init_params(m.spec, seed=3)
    x = ovae.synthetic_slices(1, hh, hh, seed=0)
    rng = np.random.default_rng(1)
    z = rng.standard_normal((1, zd)).astype(np.float32); alpha = rng.uniform(0, 1, (1, 1)).astype(np.float32)
    t0 = time.perf_counter()
    iters = 0
    while iters < 1 or (time.perf_counter() - t0 < 10.0 and iters < 64):      # ~10 s of CPU work
        m.gen_phase(p, z)
        for _ in range(5):
            m.disc_phase(p, x, z, alpha)
        iters += 1
    dt = time.perf_counter() - t0
    if limiter is not None:
        limiter.restore_original_limits()
    return {'value': round(iters / dt, 3), 'unit': 'slices/s', 'cores': threads, 'kind': 'port',
            'sample': f'{iters} WGAN-GP batch iteration(s) (1 generator + 5 critic phases each) at batch 1 with the numpy oracle, fp32, {dt:.1f} s '
                      f'(BLAS on {threads} of {os.cpu_count()} host CPUs; TF-CPU itself is not installable)'}


def bench_fanogan(args):
    """BASELINE.json configs[3]: f-AnoGAN 64x64 on the ResNet graph (models/fanogan_schlegl.py; --variant unified = models/fanogan.py,
    the graph north_star names).  One 'step' = one batch iteration of the reference's WGAN stage
    (trainers/fAnoGAN.py:97-130): 1 generator step + 5 critic steps (each incl. the second-order gradient-penalty backward) +
    their Adam updates; value = slices per second through that loop.  The encoder stage (izi_f) is timed beside it."""
    import torch
    import torch.distributed as dist
    from unsupervised_anomaly_detection_brain_mri_amd.gan_engine import GanEngine
    from unsupervised_anomaly_detection_brain_mri_amd.parallel import GanDataParallel
    from unsupervised_anomaly_detection_brain_mri_amd.utils.synthetic import synthetic_slices
    world = int(os.environ.get('WORLD_SIZE', '1')); rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    rehearsal = bool(os.environ.get('UAD_BENCH_REHEARSAL'))
    if os.environ.get('UAD_BENCH_REHEARSAL'):
        # several processes on ONE GPU: the fused bottleneck's groups of four sibling workgroups (uad_bott.hip) need all four resident at once; two processes'
        # kernels can hold each other's slots until the bounded exchange gives up (it reports, it does not hang).  The one-workgroup-per-sample form has no
        # inter-workgroup wait.  (One process per GPU -- the deployment this library is written for -- is unaffected.)
        os.environ.setdefault('UAD_BOTT_Q1', '1')    # single-GPU rehearsal of the multi-process path: every rank on cuda:0, gloo instead of RCCL
    if rehearsal:
        local_rank = 0
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    torch.cuda.set_device(local_rank)
    nccl1 = os.environ.get('UAD_BENCH_REHEARSAL') == 'nccl1' and world == 1      # the N > 1 code path (library-issued bucketed all-reduce) on ONE rank under RCCL
    if world > 1 or nccl1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if nccl1:
            os.environ.setdefault('MASTER_PORT', '29613')
            dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world) if rehearsal else dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
    hh, bs, zd = args.size or (128 if args.variant == 'anovaegan' else 64), BATCH, 128
    eng = GanEngine(hh, hh, 1, hh // 8 if args.variant == 'resnet' else 8, zd, max_batch=bs, device=f'cuda:{local_rank}', math=args.math,
                    variant=args.variant)
    graph = {'resnet': 'models/fanogan_schlegl.py (ResNet generator / critic, dim 64)', 'unified': 'models/fanogan.py (unified graph)',
             'anovaegan': 'models/anovaegan.py (AnoVAE-GAN on the unified blocks)'}[args.variant]
    rng = np.random.default_rng(3)
    flat = np.zeros(eng.nparams, np.float32)
    for name, shape, off in eng.spec:
        cnt = int(np.prod(shape))
        if name.endswith('kernel'):
            rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
            lim = np.sqrt(6.0 / (shape[-2] * rf + shape[-1] * rf))
            flat[off:off + cnt] = rng.uniform(-lim, lim, cnt)
        elif name.endswith('gamma'):
            flat[off:off + cnt] = 1.0
    eng.set_params(flat)
    dp = GanDataParallel(eng, world, library_allreduce=True, force_collectives=True) if nccl1 else GanDataParallel(eng, world)
    dp.broadcast_params(0)
    x = torch.from_numpy(synthetic_slices(bs, hh, hh, seed=1000 + rank)).cuda()
    g = torch.Generator(device='cuda').manual_seed(1 + rank)
    zs = [torch.randn(bs, zd, device='cuda', generator=g) for _ in range(6)]
    al = [torch.rand(bs, device='cuda', generator=g) for _ in range(5)]
    lr = 1e-4

    av = args.variant == 'anovaegan'

    def wgan_step():
        if av:      # trainers/AnoVAEGAN.py:97-150: VAE step, generator step, 5 critic steps
            dp.train_phase('Encoder', lr, x=x, eps=zs[0], want_images=False)
            dp.train_phase('Generator', lr, x=x, eps=zs[0], want_images=False)
            for k in range(5):
                out = dp.train_phase('Discriminator', lr, x=x, eps=zs[k + 1], alpha=al[k], want_images=False)
            return out
        dp.train_phase('Generator', lr, z=zs[0], want_images=False)
        for k in range(5):
            out = dp.train_phase('Discriminator', lr, x=x, z=zs[k + 1], alpha=al[k], want_images=False)
        return out

    def enc_step():
        if av:
            return dp.train_phase('Encoder', lr, x=x, eps=zs[0], want_images=False)
        return dp.train_phase('Encoder', lr, x=x, want_images=False)

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device='cuda', dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out

    dt, out = timed(wgan_step, args.steps, args.warmup)
    assert bool(torch.isfinite(out['disc_loss']))
    dt_e, out_e = timed(enc_step, args.steps, max(args.warmup, 1))
    assert bool(torch.isfinite(out_e['enc_loss']))
    ar_rehearsal = None
    if nccl1:
        # the same iterations WITHOUT the collectives (plain phases): the one-rank price of the data-parallel path (bucket events, the collective stream, RCCL's
        # own launches of a one-rank all-reduce)
        in_phase = dp.in_phase
        dp.close()
        dp = GanDataParallel(eng, 1, library_allreduce=False)
        dt_p, _ = timed(wgan_step, args.steps, 1)
        ar_rehearsal = {'backend': 'nccl (RCCL, one rank: code-path rehearsal)', 'library_issued_in_phase': bool(in_phase),
                        'ms_per_step_data_parallel_path': round(dt / args.steps * 1e3, 3), 'ms_per_step_plain': round(dt_p / args.steps * 1e3, 3),
                        'ratio': round(dt / dt_p, 4), 'buckets_per_phase': 'up to 4, on tensor boundaries, issued per residual block (uad_gan_allreduce_attach)'}
    if rank == 0:
        value = bs * world * args.steps / dt
        name = 'AnoVAE-GAN batch iteration (1 VAE + 1 G + 5 D steps' if av else f'f-AnoGAN ({args.variant}) WGAN-GP batch iteration (1 G + 5 D steps'
        res = {'metric': f'MRI slices/sec {name}, {hh}x{hh}, bs={bs}/GPU)',
               'value': round(value, 2), 'unit': 'slices/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
               'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
               # ResNet graph (round 4): bf16x3 = the k3 contractions on the bf16x3 spatial kernels, the penalty-value passes and the generator /
               # encoder phases on their fp32-grade three-plane form (bf16x6) -- parity-rated at 1e-4; bf16x3_all = every contraction in bf16x3 (not rated)
               'dtype': args.math,
               'data': 'synthetic',
               'config': {'workload': f'BASELINE.json configs[3]: f-AnoGAN {graph} {hh}x{hh}x1, zDim {zd}, '
                                      f'{bs} slices per GPU; step = ' + ('1 VAE + 1 generator + 5 critic phases with Adam (trainers/AnoVAEGAN.py:97-150)' if av else '1 generator + 5 critic phases with Adam (trainers/fAnoGAN.py:97-130)'),
                          'encoder_stage_ms_per_step': round(dt_e / args.steps * 1e3, 3),
                          'encoder_stage_slices_per_s': round(bs * world * args.steps / dt_e, 2), 'parallelism': f'dp{world}'}}
        if ar_rehearsal:
            res['allreduce'] = ar_rehearsal
        e_m, g_m, d_m = gan_macs('unified' if av else args.variant, hh, zd)
        # per WGAN iteration and sample: G step = G fwd + bwd (3 passes) + critic fwd + data gradient; each of the 5 critic steps =
        # G fwd + critic fwd of 3 samples, input gradient, its adjoint, data gradient of 3 and filter gradients of 4 sample-slots
        macs = (3 * g_m + 2 * d_m) + 5 * (g_m + 12 * d_m)
        if av:       # the encoder runs in front of every generator pass; plus the VAE step (E and G forward + backward)
            macs += 7 * e_m + 3 * (e_m + g_m)
        tfl = 2.0 * macs * value / 1e12
        peak = 157.3 if args.math == 'f32' else 2500.0 / 3.0
        whole = {'bound': 'mfma', 'kernel': 'whole WGAN-GP iteration', 'achieved': round(tfl, 2), 'peak': round(peak, 1),
                 'unit': 'TFLOP/s', 'frac': round(tfl / peak, 4), 'traffic': None,
                 'note': 'algorithmic FLOP of the iteration / wall time; peak = fp32 MFMA (f32 mode) or the bf16 matrix peak / 3 products per fp32 product'}
        res['roofline'] = whole
        if args.variant == 'resnet' and args.math != 'f32':
            # dominant kernel of the iteration, measured live: HIP events around every k3 launch (uad_k3_profile_*) over two more iterations
            import ctypes as C
            from unsupervised_anomaly_detection_brain_mri_amd import _lib
            lib = _lib.load()
            lib.uad_k3_profile_enable(1)
            wgan_step(); wgan_step()
            need = lib.uad_k3_profile_read(None, 0)
            buf = C.create_string_buffer(need + 16)
            lib.uad_k3_profile_read(buf, need + 16)
            lib.uad_k3_profile_enable(0)
            rows = []
            for ln in buf.value.decode().splitlines():
                v = ln.split()
                kind, p1, p2, ntaps, npl, N, MH, MW, CA, Nn, calls = map(int, v[:11])
                rows.append(dict(kind=kind, p1=p1, p2=p2, ntaps=ntaps, planes=npl, N=N, MH=MH, MW=MW, CA=CA, Nn=Nn, calls=calls, total_ms=float(v[11]),
                                 form=int(v[12]) if len(v) > 12 else 0))
            if rows:
                # group by kernel INSTANCE (kind, stride, halo / cb blocks, taps, planes): the group with the largest summed time is the dominant kernel
                groups = {}
                for r in rows:
                    groups.setdefault((r['kind'], r['p1'], r['p2'], r['ntaps'], r['planes'], r['form']), []).append(r)
                gkey, grp = max(groups.items(), key=lambda kv: sum(r['total_ms'] for r in kv[1]))
                top = max(grp, key=lambda r: r['total_ms'])                       # its most expensive launch shape
                flop = 2.0 * top['N'] * top['MH'] * top['MW'] * top['ntaps'] * top['CA'] * top['Nn']
                avg_ms = top['total_ms'] / top['calls']
                ach = flop / (avg_ms * 1e-3) / 1e12
                pk = 2500.0 / (3.0 if top['planes'] == 2 else 6.0)
                # kernel instance by form (csrc/uad_convk16.inc): tap-list 0 first kernel | 1 convk16p (pipelined) | 2 / 3 convk16q with 64 / 128 output channels per
                # workgroup; filter gradient 0 first kernel | 1 convk_w16p (pipelined, twelve waves)
                tiles = (top['MH'] // 8) * (top['MW'] // 8)
                if top['kind'] == 0:
                    tp = f"{top['p1']}, {top['p2']}, {top['ntaps']}, {top['planes']}"
                    kn, wgs, thr, what = {0: (f"convk16_kernel<8, 8, 32, 2, 2, {tp},", top['Nn'] // 64, 256, 'first form'),
                                          1: (f"convk16p_kernel<8, 8, 32, 2, 2, {tp}>", top['Nn'] // 64, 256, 'pipelined, 32 x 32 wave tiles'),
                                          2: (f"convk16q_kernel<8, 8, 32, 2, {tp}>", top['Nn'] // 64, 128, 'pipelined, 64 x 32 wave tiles, 64 channels per workgroup'),
                                          3: (f"convk16q_kernel<8, 8, 32, 4, {tp}>", top['Nn'] // 128, 256, 'pipelined, 64 x 32 wave tiles, 128 channels per workgroup')}[top['form']]
                    name = f"{kn.rstrip(',')} (tap-list k3 kernel, input stride {top['p1']}; {what})"
                    gt = str(tiles * top['N'] * wgs * thr)
                    ain = top['N'] * (top['p1'] * top['MH']) * (top['p1'] * top['MW']) * top['CA'] * 4
                    aout = top['N'] * top['MH'] * top['MW'] * top['Nn'] * 4
                    reread = {0: 'every 64-channel output block of a tile is its own workgroup and re-reads the input tile and its weight slice',
                              1: 'every 64-channel output block of a tile is its own workgroup and re-reads the input tile and its weight slice',
                              2: 'every 64-channel output block of a tile is its own workgroup and re-reads the input tile and its weight slice',
                              3: 'a workgroup owns 128 output channels of a tile: the input tile is read once per 128 channels'}[top['form']]
                else:
                    kn = f"convk_w16p_kernel<{top['p1']}, {8 if top['p1'] == 1 else 4}>" if top['form'] == 1 else f"convk_w16_kernel<{top['p1']}, {top['p2']}>"
                    name = f"{kn} (k3 filter gradient, stride {top['p1']}" + ('; pipelined, twelve waves)' if top['form'] == 1 else ')')
                    gt = None
                    ain = top['N'] * top['MH'] * top['MW'] * (top['p1'] ** 2 * top['CA'] + top['Nn']) * 4
                    aout = 9 * top['CA'] * top['Nn'] * 4
                    reread = 'every (64 x 64)-channel block re-reads its operand slices; the slabs of the split launch are written and re-read by the reduction'
                abytes = ain + aout + 9 * top['CA'] * top['Nn'] * 2 * top['planes'] * (1 if top['kind'] == 0 else 0)
                # HBM bytes per launch of that (kernel, grid) and the profiler's own average duration, from the committed rocprofv3 passes of THIS command
                # (tools/final_round6.sh -> profiles/r06_evidence_fanogan_resnet64.json, tools/evidence.py format)
                traffic = rocprof = None
                try:
                    tdoc = json.load(open(os.path.join(ROOT, 'profiles', 'r06_evidence_fanogan_resnet64.json')))
                    srcs = (f"profiles/r06_evidence_fanogan_resnet64.json: `{tdoc.get('_command')}` under rocprofv3 at commit {tdoc.get('_commit')}; not re-measured in this run")
                    trows = [r for r in tdoc.get('traffic', []) if kn in r['name'] and (gt is None or str(r['grid']) == gt)]
                    if trows:
                        r = max(trows, key=lambda r: r['fetch_bytes'] + r['write_bytes'])
                        traffic = {'bytes': int(r['fetch_bytes'] + r['write_bytes']), 'fetch_bytes': int(r['fetch_bytes']), 'write_bytes': int(r['write_bytes']),
                                   'launches_averaged': r['launches'],
                                   'source': srcs + '; --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH x2 (gfx950 correction).  Above the algorithmic bytes: ' + reread}
                    srows = [r for r in tdoc.get('stats', []) if kn in r['name']]
                    if srows:
                        r = max(srows, key=lambda r: r['total_ns'])
                        rocprof = {'avg_launch_ms_all_shapes': round(r['avg_ns'] * 1e-6, 4), 'calls': r['calls'],
                                   'source': srcs + '; --kernel-trace --stats (the summary row of a kernel template averages all its launch shapes: the per-shape time is the '
                                             'HIP-event figure above)'}
                except Exception:
                    traffic = rocprof = None
                res['roofline'] = {'bound': 'mfma', 'kernel': name,
                                   'shape': {k: top[k] for k in ('N', 'MH', 'MW', 'CA', 'Nn', 'ntaps', 'planes')},
                                   'achieved': round(ach, 2), 'peak': round(pk, 1), 'unit': 'TFLOP/s', 'frac': round(ach / pk, 4),
                                   'mfma_fraction': round(ach / pk, 4), 'hbm_fraction': round(abytes / (avg_ms * 1e-3) / 8e12, 4),
                                   'algorithmic_flop_per_launch': flop, 'algorithmic_bytes_per_launch': abytes, 'avg_launch_ms': round(avg_ms, 4),
                                   'launches_timed': top['calls'], 'group_share_of_k3_time': round(sum(r['total_ms'] for r in grp) / sum(r['total_ms'] for r in rows), 3),
                                   'instruction': f"{3 if top['planes'] == 2 else 6} x v_mfma_f32_32x32x16_bf16 per fp32 product; peak = dense bf16 MFMA 2500 TFLOP/s / products",
                                   'traffic': traffic,
                                   'rocprof': rocprof,
                                   'whole_iteration': whole}
                res['k3_kernels'] = sorted(({'kind': 'FD' if r['kind'] == 0 else 'W', 'form': r['form'], 'stride': r['p1'], 'ntaps': r['ntaps'], 'planes': r['planes'],
                                             'N': r['N'], 'grid': [r['MH'], r['MW']], 'CA': r['CA'], 'Nn': r['Nn'], 'calls': r['calls'],
                                             'avg_ms': round(r['total_ms'] / r['calls'], 4),
                                             'tflops': round(2.0 * r['N'] * r['MH'] * r['MW'] * r['ntaps'] * r['CA'] * r['Nn'] / (r['total_ms'] / r['calls'] * 1e-3) / 1e12, 1)}
                                            for r in rows), key=lambda r: -r['avg_ms'] * r['calls'])[:24]
        if not args.no_cpu_baseline and not av:
            res['cpu_baseline'] = gan_cpu_baseline(args.variant, hh, zd)
        emit_json(res)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--rounds', type=int, default=5, help='timed regions of --steps steps each; the MEDIAN round is reported (ms_per_step, value)')
    ap.add_argument('--quick', action='store_true', help='A/B runs: skip the CPU baseline, the other math mode and the trainer-loop leg')
    ap.add_argument('--math', default=os.environ.get('UAD_BENCH_MATH', 'bf16x3'), choices=['f32', 'bf16x3', 'bf16x6', 'bf16x3_all'],
                    help='bf16x3 (default): split-bf16 products on the bf16 matrix cores, fp32 accumulate, parity 1e-4 vs the '
                         'fp32 oracle; f32: exact fp32 MFMA')
    ap.add_argument('--restore-steps', type=int, default=150, help='GMVAE_spatial: restoration iterations per slice')
    ap.add_argument('--volume', action='store_true', help='GMVAE_spatial: the full-volume line (one [110,256,256] patient per GPU through restoration + post-processing + scoring)')
    ap.add_argument('--volume-slices', type=int, default=110)
    ap.add_argument('--arch', default='VAE', choices=['VAE', 'ceVAE', 'GMVAE_spatial', 'fAnoGAN'],
                    help='VAE = the headline workload (BASELINE.json configs[1]); ceVAE = configs[3] (16 slices per GPU: both '
                         'branches + the input-gradient anomaly map every step), reported for the record')
    ap.add_argument('--size', type=int, default=0, help='fAnoGAN: slice edge (default 64)')
    ap.add_argument('--variant', default='resnet', choices=['resnet', 'unified', 'anovaegan'],
                    help='fAnoGAN graph: resnet = models/fanogan_schlegl.py (the one BASELINE.json configs[3] names), unified = models/fanogan.py')
    ap.add_argument('--batch', type=int, default=0, help='slices per GPU (default 64 for VAE, 16 for ceVAE)')
    args = ap.parse_args()
    claim_stdout()
    global BATCH
    BATCH = args.batch or (64 if args.arch in ('VAE', 'fAnoGAN') else 16)
    if args.arch == 'fAnoGAN' and args.variant == 'resnet' and not args.batch:
        BATCH = 32
    cevae = args.arch == 'ceVAE'
    if args.arch == 'GMVAE_spatial':
        return bench_gmvae_volume(args) if args.volume else bench_gmvae(args)
    if args.arch == 'fAnoGAN':
        return bench_fanogan(args)

    import torch
    import torch.distributed as dist
    from unsupervised_anomaly_detection_brain_mri_amd import _lib
    from unsupervised_anomaly_detection_brain_mri_amd.engine import Engine
    from unsupervised_anomaly_detection_brain_mri_amd.parallel import DataParallelStep
    from unsupervised_anomaly_detection_brain_mri_amd.utils.synthetic import synthetic_slices

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # UAD_BENCH_REHEARSAL=nccl1 (with --gpus 1 under torch.distributed.run --nproc-per-node 1): this file's N > 1 code path -- process-group init from the launcher's
    # environment, barriers, the MAX-reduced clock, the segmented backward with one async all-reduce per bucket, the no-all-reduce leg, the stand-alone segment
    # timings -- under backend nccl (= RCCL) on the ONE rank a one-GPU box has.  Any other value: several ranks on cuda:0 over gloo.
    nccl1 = os.environ.get('UAD_BENCH_REHEARSAL') == 'nccl1'
    rehearsal = bool(os.environ.get('UAD_BENCH_REHEARSAL')) and not nccl1
    if rehearsal:
        # several processes on ONE GPU: the fused bottleneck's groups of four sibling workgroups (uad_bott.hip) need all four resident at once; two processes'
        # kernels can hold each other's slots until the bounded exchange gives up (it reports, it does not hang).  The one-workgroup-per-sample form has no
        # inter-workgroup wait.  (One process per GPU -- the deployment this library is written for -- is unaffected.)
        os.environ.setdefault('UAD_BOTT_Q1', '1')    # single-GPU rehearsal of the multi-process path: every rank on cuda:0, gloo instead of RCCL
    if rehearsal:
        local_rank = 0
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}')
    torch.cuda.set_device(local_rank)
    multi = world > 1 or nccl1
    if multi:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo', rank=rank, world_size=world) if rehearsal else dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))

    eng = Engine(args.arch, H, W, 1, INTER, ZDIM, max_batch=BATCH, device=f'cuda:{local_rank}', math=args.math)
    # identical glorot-uniform init on every rank (seed 3), zero bias, gamma 1, beta 0
    rng = np.random.default_rng(3)
    flat = np.zeros(eng.nparams, np.float32)
    for name, shape, off in eng.spec:
        cnt = int(np.prod(shape))
        if name.endswith('kernel'):
            rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
            lim = np.sqrt(6.0 / (shape[-2] * rf + shape[-1] * rf))
            flat[off:off + cnt] = rng.uniform(-lim, lim, cnt)
        elif name.endswith('gamma'):
            flat[off:off + cnt] = 1.0
    eng.set_params(flat)
    # per-rank shard of the global synthetic batch, resident in HBM
    from unsupervised_anomaly_detection_brain_mri_amd.engine import rng_fill
    x = torch.from_numpy(synthetic_slices(BATCH, H, W, seed=1000 + rank)).cuda()
    FLAT = INTER * INTER * 16
    noise_jobs = [('eps', ZDIM, 'normal', 0.0), ('mu', ZDIM, 'keep', 0.2), ('sigma', ZDIM, 'keep', 0.2), ('dec', FLAT, 'keep', 0.2)]   # rate 0.2, run.py:40
    extra = {}
    if cevae:
        x_ce = x.clone()
        x_ce[:, 40:60, 50:70] = 0          # one 20x20 context hole, the same for every slice (trainers/CE.py:130-139, A3)
        noise_jobs += [('mu_ce', ZDIM, 'keep', 0.2), ('dec_ce', FLAT, 'keep', 0.2)]
        extra = {'x_ce': x_ce}
    step_no = [0]

    def draw():
        """FRESH eps / dropout masks of this step, drawn on the device inside the timed region (one launch; the counter-based generator is
        keyed by (step, global sample index), so every rank count draws the same global batch)."""
        got = rng_fill(noise_jobs, BATCH, 1, step_no[0], rank * BATCH)
        step_no[0] += 1
        return got.pop('eps'), got
    dp = DataParallelStep(eng, world, force_collectives=True if nccl1 else None)

    def step():
        eps, masks = draw()
        return dp.train_step(x, eps, masks, lr=1e-4, beta1=0.5, want_l1=True, want_latents=False, **extra)

    def timed(steps, warmup, rounds=1):
        """`rounds` back-to-back timed regions of exactly `steps` steps each, every one bracketed by barrier + synchronize on both sides
        and reduced with MAX over the ranks; returns (seconds of the MEDIAN round, last loss, [ms per step of every round])."""
        for _ in range(warmup):
            out = step()
        per_round = []
        for _ in range(max(1, rounds)):
            torch.cuda.synchronize()
            if multi:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                out = step()
            torch.cuda.synchronize()
            if multi:
                dist.barrier()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if multi:
                t = torch.tensor([dt], device='cuda', dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            per_round.append(dt)
        dt = float(np.median(per_round))
        loss = float(out['scalars'][2].item())
        assert np.isfinite(loss) or os.environ.get('UAD_BENCH_ALLOW_NAN'), 'loss diverged'
        return dt, loss, [round(t / steps * 1e3, 4) for t in per_round]

    def profiled(math):
        """Roofline leg: per-launch-group HIP-event timing of the same step (profiling pass after the timed region)."""
        eng.profile(True)
        for _ in range(max(3, min(args.steps, 10))):
            step()
        rep = eng.profile_report()
        eng.profile(False)
        fl = flops_per_tag(BATCH * (2 if cevae else 1))
        gemm = {t: (c, ms) for t, (c, ms) in rep.items() if t in fl}
        dom = max(gemm, key=lambda t: gemm[t][1])
        dom_ms = gemm[dom][1] / gemm[dom][0]
        alg = fl[dom] / (dom_ms * 1e-3) / 1e12            # algorithmic (fp32-equivalent) TFLOP/s
        # MFMA roofline on ALGORITHMIC flops: exact-fp32 MFMA peak in f32 mode; in bf16x3 every fp32 product costs three bf16 MFMA products,
        # so the ceiling of algorithmic throughput is the dense bf16 peak / 3
        if math == 'f32':
            peak, note = PEAK_F32_MFMA_TFLOPS, 'v_mfma_f32_32x32x2_f32 (exact fp32); peak = dense fp32 MFMA'
        elif math == 'bf16x6':
            peak, note = PEAK_BF16_MFMA_TFLOPS / 6.0, ('6 x v_mfma_f32_32x32x16_bf16 per fp32 product (three bf16 planes per operand: fp32-grade results); '
                                                       'peak = dense bf16 MFMA 2500 TFLOP/s / 6 products = 2.65 x the fp32 MFMA peak')
        else:
            peak, note = PEAK_BF16_MFMA_TFLOPS / 3.0, ('3 x v_mfma_f32_32x32x16_bf16 per fp32 product; peak = dense bf16 MFMA 2500 TFLOP/s / 3 products '
                                                       '(executed bf16 FLOP/s = 3 x achieved)')
        by = bytes_per_tag(BATCH * (2 if cevae else 1), fin_bits=(math != 'f32'))
        gbs = by[dom] / (dom_ms * 1e-3) / 1e9
        traffic = None
        # the committed PMC passes and rocprofv3 summaries are of the DEFAULT command (VAE, 64 slices per launch): other workloads of this function carry none
        evidence_applies = (not cevae) and args.arch == 'VAE' and BATCH == 64
        for cand in (f'r06_traffic_{math}.json', f'r05_traffic_{math}.json', f'r04_traffic_{math}.json', f'r03_traffic_{math}.json', f'r02_traffic_{math}.json', f'r01_traffic_{math}.json') if evidence_applies else ():
            try:
                doc = json.load(open(os.path.join(ROOT, 'profiles', cand)))
                tr = doc.get(dom)
                if tr:
                    traffic = {'bytes': int(tr['fetch_bytes'] + tr['write_bytes']), 'fetch_bytes': int(tr['fetch_bytes']), 'write_bytes': int(tr['write_bytes']),
                               'source': f'profiles/{cand}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (FETCH x2 gfx950 correction), '
                                         f'measured at commit {doc.get("_commit", "(round 1 build: before the fin_bits change, dec3 kernels differ)")}; '
                                         'not re-measured in this run (PMC counters need the profiler)'}
                    break
            except Exception:
                continue
        cev_traffic, cev_rocprof = evidence_for(f'cevae_b{BATCH}', last_block_kernel(dom, len(conv_layers()) // 2, math), fl[dom], peak) if (cevae and math != 'f32') else (None, None)
        if cev_traffic:
            traffic = cev_traffic
        # the same kernel's average duration in the committed rocprofv3 --kernel-trace --stats summary of this command in this math mode (the
        # profiler's clock instead of HIP events around the launch group), with the commit it was taken at
        rocprof = None
        for tj, ks in ((f'r06_traffic_{math}.json', {'f32': 'r06_z_kernel_stats_f32.csv', 'bf16x6': 'r06_z_kernel_stats_bf16x6.csv'}.get(math, 'r06_z_kernel_stats.csv')),
                       (f'r05_traffic_{math}.json', 'r05_z_kernel_stats.csv' if math != 'f32' else 'r05_z_kernel_stats_f32.csv'),
                       (f'r04_traffic_{math}.json', 'r04_z_kernel_stats.csv' if math != 'f32' else 'r04_z_kernel_stats_f32.csv'),
                       ('r03_traffic_bf16x3.json', 'r03_z_kernel_stats.csv') if math != 'f32' else (None, None)):
            if rocprof is not None or tj is None or not evidence_applies:
                continue
            try:
                import csv
                doc = json.load(open(os.path.join(ROOT, 'profiles', tj)))
                kname = doc[dom]['kernel']
                rows = list(csv.DictReader(open(os.path.join(ROOT, 'profiles', ks))))
                cands = [row for row in rows if kname in row['Name']]
                if cands:
                    row = max(cands, key=lambda r: float(r['TotalDurationNs']))
                    avg_ms = float(row['AverageNs']) * 1e-6
                    per_step = min([int(r['Calls']) for r in rows if 'adam_kernel' in r['Name']] or [0])      # launches of a once-per-step kernel
                    shared = per_step > 0 and int(row['Calls']) > per_step
                    rocprof = {'avg_launch_ms': round(avg_ms, 4), 'calls': int(row['Calls']), 'frac': round(fl[dom] / (avg_ms * 1e-3) / 1e12 / peak, 4),
                               'source': f'profiles/{ks} (bench.py --steps 10 --warmup 3 --quick --math {math} under rocprofv3 --kernel-trace --stats), '
                                         f'commit {doc.get("_commit")}; not re-measured in this run'
                                         + ('; the summary row averages every launch of this kernel template (other layers share it)' if shared else '')}
            except Exception:
                rocprof = None
        kernels = {t: {'ms': round(ms / c, 4), 'tflops': round(fl[t] / (ms / c * 1e-3) / 1e12, 2) if t in fl else None,
                       'gbs': round(by[t] / (ms / c * 1e-3) / 1e9, 1) if t in by else None}
                   for t, (c, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1])}
        roof = {'bound': 'mfma', 'kernel': dom, 'achieved': round(alg, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
                'frac': round(alg / peak, 4), 'mfma_fraction': round(alg / peak, 4),
                'hbm_fraction': round(gbs / PEAK_HBM_GBS, 4), 'hbm_achieved_gbs': round(gbs, 1), 'hbm_peak_gbs': PEAK_HBM_GBS,
                'algorithmic_bytes_per_launch': int(by[dom]), 'algorithmic_flop_per_launch': int(fl[dom]),
                'traffic': traffic, 'avg_launch_ms': round(dom_ms, 4), 'rocprof': rocprof or cev_rocprof, 'instruction': note}
        return roof, kernels

    SPEC_CLOCK_GHZ = 2.4

    def with_clock(roof):
        """The roofline peaks are priced at the 2.4 GHz spec clock; the chip runs this kernel mix under its power limit.  Measured here, live,
        beside the same step (engine.shader_clock_under: a probe wave reads s_memtime against the 100 MHz s_memrealtime while steps run):
        `clock_ghz_measured`, and every fraction re-priced at that clock (`frac_at_measured_clock` = frac x 2.4 / clock)."""
        from unsupervised_anomaly_detection_brain_mri_amd.engine import shader_clock_under
        try:
            ghz = shader_clock_under(step, window_ms=30.0)
        except Exception as e:            # an older library build behind UAD_LIB
            roof['clock_ghz_measured'] = None
            roof['clock_note'] = f'not measured: {e}'
            return roof
        roof['clock_ghz_spec'] = SPEC_CLOCK_GHZ
        roof['clock_ghz_measured'] = round(ghz, 3)
        roof['frac_at_measured_clock'] = round(roof['frac'] * SPEC_CLOCK_GHZ / ghz, 4)
        if roof.get('rocprof'):
            roof['rocprof']['frac_at_measured_clock'] = round(roof['rocprof']['frac'] * SPEC_CLOCK_GHZ / ghz, 4)
        roof['clock_note'] = ('shader clock sustained under this step (whole-chip DVFS), sampled for 30 ms by one probe wave beside the running steps; '
                              'peak x clock / 2.4 GHz is the matrix-pipe ceiling at that clock')
        return roof

    dt, loss, round_ms = timed(args.steps, args.warmup, args.rounds)
    roof, kernels = profiled(args.math)
    roof = with_clock(roof)
    # the other math mode, same handle, for the record (rank 0 / single GPU only; not the headline value)
    other = None
    others = []
    if world == 1 and not args.quick:
        # exact fp32 (north_star's "fp32 MFMA roofline" mode) and bf16x6 (round 6: the fp32-GRADE mode on the bf16 matrix cores, 1e-5 parity) ride beside the default
        for om in [mm for mm in ('f32', 'bf16x6', 'bf16x3') if mm != args.math]:
            eng.set_math(om)
            odt, _, oround_ms = timed(args.steps, max(2, args.warmup), args.rounds)
            oroof, _ = profiled(om)
            oroof = with_clock(oroof)
            ovalue = BATCH * args.steps / odt
            rec = {'math': om, 'value': round(ovalue, 1), 'ms_per_step': round(odt / args.steps * 1e3, 4), 'round_ms_per_step': oround_ms, 'roofline': oroof}
            if om != 'bf16x3':
                # north_star: ">= 60 % fp32 MFMA roofline": algorithmic fp32 FLOP of the whole step over the fp32 MFMA peak
                rec['step_fraction_of_fp32_mfma_peak'] = round(ovalue * train_flops_per_slice() / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
            others.append(rec)
        eng.set_math(args.math)
        other = others[0] if others else None

    # per-segment gradient all-reduce, timed on its own after the timed region (N > 1): what the backward has to hide
    allreduce = None
    if multi:
        # communication the backward did NOT hide = (step with the gradient all-reduces) - (same step without them), same ranks, same run
        dp.no_allreduce = True
        ndt, _, nround_ms = timed(args.steps, 2, args.rounds)
        dp.no_allreduce = False
        exposed_ms = (dt - ndt) / args.steps * 1e3
        names = {_lib.SEG_DECODER: 'decoder', _lib.SEG_BOTTLENECK: 'bottleneck', _lib.SEG_ENCODER_HI: 'encoder_deep', _lib.SEG_ENCODER_LO: 'encoder_first_blocks'}
        allreduce = {'ranks': dist.get_world_size(), 'backend': dist.get_backend() + (' (single-GPU rehearsal)' if rehearsal else ' (RCCL, one rank: code-path rehearsal)' if nccl1 else ' (RCCL over xGMI)'),
                     'buckets': dp.buckets, 'bucket_bytes': [int(c * 4) for _, _, c in dp.plan],
                     'exposed_comm_ms': round(exposed_ms, 4), 'ms_per_step_without_allreduce': round(ndt / args.steps * 1e3, 4),
                     'round_ms_per_step_without_allreduce': nround_ms,
                     'env': {k: os.environ.get(k) for k in ('NCCL_ALGO', 'NCCL_PROTO', 'NCCL_MIN_NCHANNELS', 'NCCL_MAX_NCHANNELS', 'RCCL_MSCCL_ENABLE',
                                                            'UAD_DP_BUCKETS', 'HSA_ENABLE_IPC_MODE_LEGACY')},
                     'segments': {}}
        for seg, (off, cnt) in dp.segs.items():
            if cnt == 0:
                continue
            buf = dp.grads[off:off + cnt]
            for _ in range(3):
                dist.all_reduce(buf)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); dist.barrier()
            e0.record()
            for _ in range(10):
                dist.all_reduce(buf)
            e1.record(); torch.cuda.synchronize()
            allreduce['segments'][names[seg]] = {'bytes': int(cnt * 4), 'ms': round(e0.elapsed_time(e1) / 10, 4)}

    # the reference's unit of work is the process() loop (trainers/VAE.py:76-103): one TRAIN epoch of trainers.VAE on an HBM-resident
    # slice set -- batch gather, noise and scalar bookkeeping included -- next to the bare step above
    loop = None
    if world == 1 and not cevae and args.arch == 'VAE' and not args.quick:
        from unsupervised_anomaly_detection_brain_mri_amd.models import variational_autoencoder
        from unsupervised_anomaly_detection_brain_mri_amd.trainers import VAE, Phase
        from unsupervised_anomaly_detection_brain_mri_amd.utils.default_config_setup import get_config, get_options
        from unsupervised_anomaly_detection_brain_mri_amd.utils.slice_cache import DeviceDataset
        eng.close()
        nb = max(args.steps, 20)
        base = synthetic_slices(256, H, W, seed=77)
        imgs = np.concatenate([base] * ((BATCH * nb + 255) // 256))[:BATCH * nb]
        dsd = DeviceDataset(imgs, np.zeros(len(imgs), np.int64), seed=0)
        opt = get_options(batchsize=BATCH, learningrate=1e-4, numEpochs=1, zDim=ZDIM, outputWidth=W, outputHeight=H,
                          config={'CHECKPOINTDIR': '/tmp/uad_bench_ck', 'SAMPLEDIR': '/tmp/uad_bench_smp'})
        cfg = get_config(VAE, opt, 'ADAM', [INTER, INTER], 0.2, dsd)
        cfg.quiet = True; cfg.useTensorboard = False
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            tm = VAE(None, cfg, network=variational_autoencoder)
            tm.engine.set_math(args.math)
            tm.process(dsd, 0, Phase.TRAIN)              # warm-up epoch
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sc = tm.process(dsd, 1, Phase.TRAIN)
            torch.cuda.synchronize()
            ldt = time.perf_counter() - t0
        loop = {'value': round(BATCH * nb / ldt, 1), 'unit': 'slices/s', 'ms_per_step': round(ldt / nb * 1e3, 4), 'steps': nb,
                'what': 'trainers.VAE.process(dataset, epoch, TRAIN): DeviceDataset.next_batch gather + device noise + train step per batch, '
                        'one host synchronisation per epoch', 'epoch_loss': float(sc['loss'])}
        tm.engine.close()

    if rank == 0:
        slices = BATCH * world * args.steps
        value = slices / dt
        res = {
            'metric': 'MRI slices/sec VAE train step (128x128, bs=64)' if not cevae else
                      f'MRI slices/sec ceVAE train step (128x128, bs={BATCH}/GPU)',
            'value': round(value, 1), 'unit': 'slices/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'rounds': args.rounds, 'round_ms_per_step': round_ms,      # every round times exactly `steps` steps; value / ms_per_step = the median round
            'ms_per_step': round(dt / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.math, 'data': 'synthetic',
            'config': {'workload': ('BASELINE.json configs[1]: VAE 128x128x1 slices, batch 64 per GPU, '
                                    'fwd + bwd + TF-Adam (lr 1e-4, beta1 0.5), dropout 0.2, inter_res 8, zDim 128') if not cevae else
                                   (f'BASELINE.json configs[3]: ceVAE 128x128x1 slices, batch {BATCH} per GPU; x and the context-masked '
                                    'x_ce through shared layers (one 2n-sample pass), loss = mean(rec_vae + kl + rec_ce), '
                                    'fwd + bwd + TF-Adam + the input-gradient anomaly map of every step'),
                       'math': args.math + (' = fp32 operands split hi+lo into bf16, hi*hi+hi*lo+lo*hi on the bf16 MFMA, fp32 accumulate; '
                                            'parity 1e-4 vs the fp32 oracle (tests/test_gpu_model.py)' if args.math == 'bf16x3' else ' = exact fp32 MFMA'),
                       'global_batch': BATCH * world, 'per_gpu_batch': BATCH,
                       'parallelism': f'dp{world}' if world > 1 else 'single',
                       'step_tflops': round(value * train_flops_per_slice() * (2 if cevae else 1) / 1e12, 2),
                       'final_loss': round(loss, 4)},
            'kernels': kernels,
        }
        res['roofline'] = roof
        step_bytes = sum(bytes_per_tag(BATCH * (2 if cevae else 1), fin_bits=(args.math != 'f32')).values())
        res['roofline']['whole_step'] = {
            'algorithmic_tflops': res['config']['step_tflops'], 'mfma_fraction': round(res['config']['step_tflops'] / roof['peak'], 4),
            'algorithmic_gbytes': round(step_bytes / 1e9, 3),
            'hbm_fraction': round(step_bytes / (dt / args.steps) / 1e9 / PEAK_HBM_GBS, 4),
            'mfma_fraction_at_measured_clock': (round(res['config']['step_tflops'] / roof['peak'] * SPEC_CLOCK_GHZ / roof['clock_ghz_measured'], 4)
                                                if roof.get('clock_ghz_measured') else None),
            'note': 'conv launch groups only (the first / final single-channel kernels, the bottleneck and Adam add < 8 % of the bytes)'}
        if other:
            res['other_math_mode'] = other                       # exact fp32 (kept under the key earlier rounds used)
            res['other_math_modes'] = others                     # + bf16x6: fp32-grade numerics (1e-5 parity, tests/test_gpu_scale_parity.py) on the bf16 matrix cores
        if loop:
            res['trainer_loop_slices_per_s'] = loop['value']
            res['trainer_loop'] = loop
        if allreduce:
            res['allreduce'] = allreduce
        if world == 1 and not args.no_cpu_baseline and not args.quick and not cevae:
            res['cpu_baseline'] = cpu_baseline()
        emit_json(res)
    if multi:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
