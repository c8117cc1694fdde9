"""Slice-batch data parallelism: one process per GPU, gradients summed with RCCL all-reduce over xGMI
(torch.distributed backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests).

The reference has no distributed code; this is the build's addition.  Because BatchNorm runs
with frozen statistics (SURVEY.md §8a note 1) per-sample gradients are independent, so DP over the slice batch equals
the single-device big batch up to fp32 summation order.  The flat fp32 gradient buffer is laid out Encoder | Bottleneck |
Decoder; the backward finishes the segments in the order Decoder, Bottleneck, Encoder and each segment's all-reduce
is issued (async, on RCCL's stream) as soon as its kernels are enqueued, overlapping with the remaining backward.
"""
import os

import torch
import torch.distributed as dist

from . import _lib

# the encoder segment goes in two parts: the deep blocks' variables (2.5 MB of its 2.7 MB at the default depth) are complete two blocks before
# the backward ends, so the all-reduce that is exposed after the last kernel moves only the first two blocks' kernels (0.2 MB)
SEGMENT_ORDER = (_lib.SEG_DECODER, _lib.SEG_BOTTLENECK, _lib.SEG_ENCODER_HI, _lib.SEG_ENCODER_LO)


def _all_ok(ok, group=None):
    """True iff `ok` holds on EVERY rank of the group (one MIN all-reduce): the outcome of a step that can fail on one rank only becomes a
    collective decision, so that all ranks take the same path afterwards (library-issued vs torch.distributed all-reduce) or raise together."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return bool(ok)
    on_dev = dist.get_backend(group) == 'nccl'
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device='cuda' if on_dev else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()))


class RcclComm:
    """An RCCL communicator owned by libuad_hip.so (include/uad_hip.h: uad_rccl_*), one rank per process, created over an EXISTING torch.distributed
    group: rank 0 draws the ncclUniqueId, one broadcast over the group hands it to the others, every rank joins with ncclCommInitRank on its current
    device.  The library enqueues its all-reduces on this communicator itself -- torch's process group is only the bootstrap channel.

    Every step that can fail on one rank alone is agreed on over the group before the next collective: (1) every rank draws an id (which loads
    librccl; rank 0's is the one used) and the ranks agree that all could; (2) rank 0 ALWAYS broadcasts -- a rank that failed never leaves the others
    waiting in a broadcast that it then pairs with some later collective of a different size; (3) after ncclCommInitRank the ranks agree again, and a
    rank that succeeded where another failed destroys its communicator.  The constructor therefore raises on all ranks or on none."""

    def __init__(self, group=None):
        import ctypes as C
        self.lib = _lib.load()
        self.handle = None
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        idbuf = (C.c_ubyte * 128)()
        err = None
        try:
            _lib.check(self.lib.uad_rccl_unique_id(idbuf, 128))
        except Exception as e:      # librccl not loadable on this rank, ncclGetUniqueId refused ...
            err = e
        if not _all_ok(err is None, group):
            raise RuntimeError('RCCL is not usable on every rank' + (f' (this rank: {err})' if err else ' (another rank failed)'))
        on_dev = dist.get_backend(group) == 'nccl'
        t = torch.tensor(list(idbuf), dtype=torch.uint8, device='cuda' if on_dev else 'cpu')
        if self.world > 1:
            dist.broadcast(t, src=0, group=group)
        raw = bytes(t.cpu().tolist())
        comm = C.c_void_p()
        torch.cuda.synchronize()
        try:
            _lib.check(self.lib.uad_rccl_comm_create(raw, self.world, self.rank, C.byref(comm)))
            self.handle = comm.value
        except Exception as e:
            err = e
        if not _all_ok(err is None, group):
            self.close()
            raise RuntimeError('ncclCommInitRank did not succeed on every rank' + (f' (this rank: {err})' if err else ' (another rank failed)'))

    def allreduce_(self, tensor, stream=None):
        """In-place float32 sum over the ranks, enqueued on `stream` (default: the current stream); returns at once."""
        import ctypes as C
        assert tensor.dtype == torch.float32 and tensor.is_contiguous() and tensor.is_cuda
        st = stream if stream is not None else torch.cuda.current_stream(tensor.device).cuda_stream
        _lib.check(self.lib.uad_rccl_allreduce(C.c_void_p(self.handle), C.c_void_p(tensor.data_ptr()), tensor.numel(), C.c_void_p(st)))

    def close(self):
        if getattr(self, 'handle', None):
            torch.cuda.synchronize()
            self.lib.uad_rccl_comm_destroy(self.handle)
            self.handle = None


def _library_allreduce_default():
    """The library-issued path is the default when the process group's backend is RCCL ("nccl") and UAD_DP_LIBRARY_AR is not 0."""
    if os.environ.get('UAD_DP_LIBRARY_AR', '1') == '0':
        return False
    return dist.is_initialized() and dist.get_backend() == 'nccl'


def allreduce_segments(grads_flat, segments, world, async_op=True):
    """grads_flat: 1-D tensor (device or CPU); segments: [(offset, count)] -> list of work handles."""
    works = []
    for off, cnt in segments:
        if world > 1:
            works.append(dist.all_reduce(grads_flat[off:off + cnt], op=dist.ReduceOp.SUM, async_op=async_op))
    return works


def bucket_plan(segs, buckets):
    """Which contiguous slice of the flat gradient buffer is all-reduced after which backward segment.
    segs: {segment: (offset, count)}; buckets 4 = one all-reduce per segment (DECODER, BOTTLENECK, ENCODER_HI, ENCODER_LO: the most overlap,
    four collective latencies), 3 = Decoder | Encoder-deep + Bottleneck | first encoder blocks, 2 = everything but the first two encoder
    blocks' kernels in ONE collective issued two layers before the backward ends (6.8 MB hidden behind the last two layers) + the 0.2 MB rest,
    1 = the whole buffer after the last backward kernel (one latency, no overlap).  The buffer is laid out Encoder(LO | HI) | Bottleneck |
    Decoder, so every merged bucket is one contiguous slice.  Returns [(segment after which to issue, offset, count)]."""
    def span(parts):
        parts = [segs[p] for p in parts if segs[p][1] > 0]
        if not parts:
            return 0, 0
        lo = min(o for o, _ in parts)
        hi = max(o + c for o, c in parts)
        assert hi - lo == sum(c for _, c in parts), 'merged gradient segments must be contiguous'
        return lo, hi - lo
    if buckets == 4:
        groups = [(s, (s,)) for s in SEGMENT_ORDER]
    elif buckets == 3:
        groups = [(_lib.SEG_DECODER, (_lib.SEG_DECODER,)), (_lib.SEG_ENCODER_HI, (_lib.SEG_ENCODER_HI, _lib.SEG_BOTTLENECK)),
                  (_lib.SEG_ENCODER_LO, (_lib.SEG_ENCODER_LO,))]
    elif buckets == 2:
        groups = [(_lib.SEG_ENCODER_HI, (_lib.SEG_ENCODER_HI, _lib.SEG_BOTTLENECK, _lib.SEG_DECODER)), (_lib.SEG_ENCODER_LO, (_lib.SEG_ENCODER_LO,))]
    elif buckets == 1:
        groups = [(_lib.SEG_ENCODER_LO, SEGMENT_ORDER)]
    else:
        raise ValueError('UAD_DP_BUCKETS must be 4, 3, 2 or 1')
    return [(after,) + span(parts) for after, parts in groups]


class DataParallelStep:
    """train_step() for rank-local shards; equal per-rank batch sizes are assumed (weak scaling).
    buckets (default: env UAD_DP_BUCKETS, else 4): how the four gradient segments are merged into all-reduce calls (bucket_plan).
    no_allreduce (default: env UAD_DP_NO_ALLREDUCE): skip the collectives -- the step then trains on the local gradient; bench.py uses the
    difference of the two step times as the communication the backward did not hide.
    force_collectives (default: env UAD_DP_FORCE_COLLECTIVES): take the segmented backward + per-bucket all-reduce path even at world 1 (an
    all-reduce over one rank is the identity, so the step must end on the bits of the plain one): lets a ONE-GPU box run RCCL on the
    engine-owned gradient view (tests/test_gpu_dp_nccl.py) before the 8-GPU node does.
    Deferred joins (round 4; env UAD_DP_NO_DEFER=1 turns them off): a segment's parameter gradients are written on the engine's side stream, and
    `uad_backward(segment)` makes the compute stream wait for that stream before it returns -- three extra stalls per step, each exposing the tail of
    a slab reduction (measured with RCCL on one rank: 1.07 ms per step against 0.86 for the unsegmented backward).  Only the collective needs that
    order: `Engine.backward_deferred` skips the wait and names the stream the slice is complete in, the all-reduce is issued under THAT stream (the
    process group's stream then waits for it, not the compute stream), and the compute stream goes straight on with the next segment."""

    def __init__(self, engine, world=None, buckets=None, no_allreduce=None, force_collectives=None, library_allreduce=None, comm=None):
        self.eng = engine
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.force = bool(int(os.environ.get('UAD_DP_FORCE_COLLECTIVES', '0'))) if force_collectives is None else bool(force_collectives)
        if self.force and not dist.is_initialized():
            raise RuntimeError('force_collectives needs an initialised process group')
        self.grads = engine.buffer(_lib.BUF_GRADS) if (self.world > 1 or self.force) else None
        self.segs = {s: engine.grad_segment(s) for s in SEGMENT_ORDER}
        self.buckets = int(buckets if buckets is not None else os.environ.get('UAD_DP_BUCKETS', '4'))
        self.plan = bucket_plan(self.segs, self.buckets)
        self.no_allreduce = bool(int(os.environ.get('UAD_DP_NO_ALLREDUCE', '0'))) if no_allreduce is None else bool(no_allreduce)
        self.defer = not bool(int(os.environ.get('UAD_DP_NO_DEFER', '0'))) and hasattr(engine, 'backward_deferred')
        # library-issued all-reduce (round 5; uad_allreduce_attach): the engine enqueues ncclAllReduce itself right behind each bucket's slab
        # reductions -- no torch process-group hand-off per collective.  Needs RCCL (backend "nccl") and an engine with the export; the
        # torch.distributed path below stays as the fallback (gloo CPU tests, UAD_DP_LIBRARY_AR=0).
        want_lib = _library_allreduce_default() if library_allreduce is None else bool(library_allreduce)
        self.comm = None
        self._own_comm = False
        if want_lib and (self.world > 1 or self.force) and hasattr(engine, 'allreduce_attach'):
            # Whether the library path is taken is a COLLECTIVE decision: RcclComm() raises on all ranks or on none, and the attach
            # -- which can fail on one rank alone -- is agreed on before anybody uses it: a rank on the torch process group beside ranks issuing
            # ncclAllReduce on the library's communicator would deadlock the first train_step.
            err = None
            try:
                self.comm = comm if comm is not None else RcclComm()      # (comm=: a communicator the caller created earlier)
                self._own_comm = comm is None
            except Exception as e:      # librccl not loadable, communicator creation refused ...: on every rank (RcclComm agrees before it raises)
                err = e
            if err is None:
                try:
                    engine.allreduce_attach(self.comm, self.world, self.plan)
                except Exception as e:
                    err = e
            if not _all_ok(err is None):
                self._drop_comm()
                if library_allreduce:   # asked for explicitly: do not hide the failure (raised on every rank)
                    raise RuntimeError(f'library-issued all-reduce unavailable: {err if err else "another rank failed"}')
                import sys
                print(f'uad: library-issued all-reduce unavailable ({err if err else "another rank failed"}); every rank falls back to torch.distributed', file=sys.stderr)
        if self.comm is None and hasattr(engine, 'allreduce_attach') and getattr(engine, '_ar_comm', None) is not None:
            engine.allreduce_attach(None, 1, None)      # a communicator an earlier DataParallelStep left attached must not outlive the choice of the torch path
            engine._ar_comm = None
        if self.comm is None and (self.world > 1 or self.force) and dist.is_initialized() and dist.get_backend() == 'nccl' and getattr(engine, 'created_before_process_group', False):
            # torch path under RCCL: with the handle created BEFORE the communicator the process group's stream lands on a hardware queue it shares
            # with a stream it waits for -- every all-reduce then costs ~0.2 ms of stall (measured).  Enforced, not only documented.
            raise RuntimeError('DataParallelStep over torch.distributed/nccl: create the process group (init_process_group(..., device_id=...)) BEFORE the '
                               'engine, or use the library-issued all-reduce (UAD_DP_LIBRARY_AR=1, the default)')

    def _drop_comm(self):
        if self.comm is not None:
            try:
                if getattr(self.eng, 'handle', None) and hasattr(self.eng, 'allreduce_attach'):
                    self.eng.allreduce_attach(None, 1, None)
                    self.eng._ar_comm = None
            finally:
                if self._own_comm:
                    self.comm.close()
                self.comm = None

    def close(self):
        """Detaches the library-issued all-reduce from the engine and destroys the communicator this object created (one per DataParallelStep: re-creating
        the step on the same engine without close() leaked one communicator each time)."""
        self._drop_comm()

    def broadcast_params(self, src=0):
        if self.world > 1 or self.force:
            dist.broadcast(self.eng.buffer(_lib.BUF_PARAMS), src=src)

    def train_step(self, x, eps=None, masks=None, lr=1e-4, beta1=0.5, beta2=0.999, adam_eps=1e-8, **kw):
        eng = self.eng
        out = eng.forward(x, eps, masks, want_backward=True, **kw)
        if self.world == 1 and not self.force:
            eng.backward(_lib.SEG_ALL)
            eng.adam_step(lr, beta1, beta2, adam_eps, 1.0)
            return out
        if self.comm is not None and not self.no_allreduce:
            # library-issued: each call runs one backward segment and enqueues the all-reduce of the buckets it completes; after the last one the
            # compute stream has been ordered behind every bucket
            for seg in SEGMENT_ORDER:
                eng.backward_allreduce(seg)
            eng.adam_step(lr, beta1, beta2, adam_eps, 1.0 / self.world)
            return out
        works = []
        issue = {after: (off, cnt) for after, off, cnt in self.plan}
        for seg in SEGMENT_ORDER:
            ready = eng.backward_deferred(seg) if self.defer else eng.backward(seg)      # a stream, or None = the current one
            off, cnt = issue.get(seg, (0, 0))
            if cnt > 0 and not self.no_allreduce:          # (the spatial AE has no bottleneck variables)
                if ready is not None:
                    with torch.cuda.stream(ready):         # the collective is ordered behind the side stream's reductions, the compute stream is not
                        works.append(dist.all_reduce(self.grads[off:off + cnt], op=dist.ReduceOp.SUM, async_op=True))
                else:
                    works.append(dist.all_reduce(self.grads[off:off + cnt], op=dist.ReduceOp.SUM, async_op=True))
        for w in works:
            w.wait()
        # local grads are d(mean over the local batch); sum / world = d(mean over the global batch)
        eng.adam_step(lr, beta1, beta2, adam_eps, 1.0 / self.world)
        return out

    def allreduce_scalars(self, scalars):
        if self.world > 1 or self.force:
            dist.all_reduce(scalars, op=dist.ReduceOp.SUM)
            scalars /= self.world
        return scalars


class GanDataParallel:
    """f-AnoGAN phases under slice-batch DP: every phase loss is a mean over the local batch of per-sample terms (the critic's
    LayerNorm is per sample, the penalty per sample and column), so the all-reduced sum / world of the trained group's gradient
    slice equals the big-batch gradient.  One all-reduce per phase, over that group's slice only (Encoder 1.2 M, Generator
    1.5 M, Discriminator 0.7 M floats at 128x128)."""

    def __init__(self, engine, world=None, library_allreduce=None, comm=None, force_collectives=False):
        """comm: a communicator the caller created (tests: a stand-in librccl); force_collectives: take the collective path at world 1 too."""
        self.eng = engine
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.grads = engine.buffer(_lib.BUF_GRADS) if self.world > 1 else None
        # library-issued RCCL (round 5): the group's slice is all-reduced ON THE PHASE'S STREAM by libuad_hip.so's own communicator -- stream-ordered
        # between the phase's last gradient kernel and the group's Adam launch, the host does not block and no process-group stream is involved.
        # (The phases of a WGAN iteration form a dependency chain -- every forward reads the parameters the previous phase's Adam wrote -- so there is
        # no later work the collective could legally overlap with; what the library path removes is the blocking host wait and the event hand-off.)
        want_lib = _library_allreduce_default() if library_allreduce is None else bool(library_allreduce)
        self.comm = comm
        self._own_comm = False
        self.in_phase = False          # True: uad_gan_phase all-reduces the trained group itself, in buckets overlapped with its backward (round 6)
        self.force = bool(force_collectives)
        if self.force and self.grads is None:
            self.grads = engine.buffer(_lib.BUF_GRADS)
        if want_lib and (self.world > 1 or self.force):
            err = None
            try:
                if self.comm is None:
                    self.comm = RcclComm()      # raises on every rank or on none (it agrees over the group first)
                    self._own_comm = True
            except Exception as e:
                err = e
            if err is None and hasattr(engine, 'allreduce_attach') and getattr(engine, 'variant', '') != 'aae':
                try:
                    engine.allreduce_attach(self.comm, self.world)
                    self.in_phase = True
                except Exception as e:
                    err = e
            if not _all_ok(err is None):        # (the attach can fail on one rank alone: agree before anybody relies on it)
                if self.in_phase:
                    engine.allreduce_attach(None, 1)
                self.in_phase = False
                if self._own_comm and self.comm is not None:
                    self.comm.close()
                self.comm = None
                if library_allreduce:
                    raise RuntimeError(f'library-issued all-reduce unavailable: {err if err else "another rank failed"}')
                import sys
                print(f'uad: library-issued all-reduce unavailable ({err if err else "another rank failed"}); every rank falls back to torch.distributed', file=sys.stderr)

    def close(self):
        if self.in_phase and getattr(self.eng, 'handle', None):
            self.eng.allreduce_attach(None, 1)
        self.in_phase = False
        if self.comm is not None and self._own_comm:
            self.comm.close()
        self.comm = None

    def broadcast_params(self, src=0):
        if self.world > 1 and dist.is_initialized():
            dist.broadcast(self.eng.buffer(_lib.BUF_PARAMS), src=src)

    def train_phase(self, group, lr, beta1=0.5, beta2=0.9, adam_eps=1e-8, **kw):
        out = self.eng.phase(group, want_backward=True, **kw)
        if self.in_phase:
            pass          # the library enqueued the group's bucketed all-reduce behind (and beside) the phase's own backward kernels
        elif self.world > 1 or self.force:
            # AnoVAE-GAN's 'Encoder' phase is optim_vae: Encoder + Generator variables (one contiguous slice)
            red = 'VAE' if (getattr(self.eng, 'variant', '') == 'anovaegan' and group == 'Encoder') else group
            off, cnt = self.eng.group(red)
            if self.comm is not None:
                self.comm.allreduce_(self.grads[off:off + cnt])
            else:
                dist.all_reduce(self.grads[off:off + cnt], op=dist.ReduceOp.SUM)
        self.eng.adam(group, lr, beta1, beta2, adam_eps, 1.0 / self.world)
        return out

    def allreduce_scalars(self, scalars):
        if self.world > 1:
            dist.all_reduce(scalars, op=dist.ReduceOp.SUM)
            scalars /= self.world
        return scalars
