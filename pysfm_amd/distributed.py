"""Multi-GPU bundle adjustment: one process per GPU, tracks sharded, ONE collective.

Every term of the reduced camera system is a sum over points
(bundle_adjuster.py:230-234, 263-276):

    S = diag(HCC) - sum_k W_k HPP_k^-1 W_k^T ,   b = bC - sum_k W_k HPP_k^-1 bP_k ,
    HCC_i = sum_{k seen by i} Jc^T Jc ,          bC_i = sum Jc^T r .

So each rank owns a contiguous range of tracks (with all their observations, HPP, bP,
W) and ALL cameras (96 B each, replicated), forms its partial (S_g, b_g) and a single
``all_reduce(sum, fp64)`` over RCCL/xGMI of the [S | b] buffer yields the full system
on every rank.  The reduced solve is replicated, back-substitution and the point
update are local; the trial's record (cost partials + two status words, 16 KB) is one more
small all-reduce.  There is no other data-path communication.

Usage (torchrun, one rank per GPU):

    comm = ShardComm()                                    # wraps torch.distributed
    ids = shard_tracks(bundle, comm.rank, comm.world_size)
    ba = BundleAdjuster(device=comm.local_rank, comm=comm)
    ba.set_bundle(bundle, track_ids=ids)
    ba.optimize()
"""
import contextlib
import os
import sys

import numpy as np


def shard_bounds(track_lengths, world_size):
    """Split tracks 0..nt-1 into `world_size` contiguous ranges with balanced
    observation counts.  Returns world_size+1 boundaries."""
    L = np.asarray(track_lengths, np.int64)
    nt = len(L)
    csum = np.concatenate(([0], np.cumsum(L)))
    total = csum[-1]
    bounds = [0]
    for g in range(1, world_size):
        target = total * g / float(world_size)
        k = int(np.searchsorted(csum, target, side='left'))
        k = min(max(k, bounds[-1]), nt)
        bounds.append(k)
    bounds.append(nt)
    return bounds


def tree_cut(bundle, world_size, plan):
    """Where the reduced solve spread over the ranks (csrc/ba_dist.h) wants the tracks cut: `plan(nco, hb, world)` is
    HipBackend.dist_plan; assumes the adjuster's default selection (every camera but the first optimised: position =
    camera index - 1).  Returns the camera INDICES at which rank r's tracks start (length world + 1), or None."""
    cam, trk, _ = bundle.observation_table()
    nc, nt = len(bundle.cameras), len(bundle.tracks)
    if nc < 3 or nt == 0:
        return None
    pos = np.asarray(cam, np.int64) - 1
    ok = pos >= 0
    lo = np.full(nt, np.iinfo(np.int64).max, np.int64)
    hi = np.full(nt, -1, np.int64)
    np.minimum.at(lo, np.asarray(trk)[ok], pos[ok])
    np.maximum.at(hi, np.asarray(trk)[ok], pos[ok])
    hb = int(np.max(np.where(hi >= 0, hi - lo, 0)))
    cut = plan(nc - 1, max(hb, 1), world_size)
    if cut is None:
        return None
    cb, N, P = cut
    return [0] + [min(nc, r * P * cb + 1) for r in range(1, world_size)] + [nc]


def shard_tracks(bundle, rank, world_size, plan=None):
    """Track ids owned by `rank`: the tracks ordered by their first camera (the caller's order when that is
    already ascending), cut into `world_size` consecutive ranges - a rank's partial reduced system then covers one
    stretch of the band instead of all of it, whatever order the tracks come in.  Where the cuts fall: with `plan`
    (HipBackend.dist_plan) and a scene the distributed reduced solve applies to, at the camera positions its elimination
    tree is cut at (tree_cut); otherwise balanced by observation count.  A rank may get no tracks when there are more
    ranks than usable tracks."""
    cam, trk, _ = bundle.observation_table()
    nt = len(bundle.tracks)
    L = np.bincount(trk, minlength=nt)
    first = np.full(nt, np.iinfo(np.int64).max, np.int64)
    np.minimum.at(first, trk, cam)
    order = np.arange(nt) if np.all(np.diff(first) >= 0) else np.argsort(first, kind='stable')
    cuts = tree_cut(bundle, world_size, plan) if plan is not None and world_size > 1 else None
    if cuts is not None:
        # a track belongs to the rank whose interval holds its first OPTIMISED camera (camera 0, the frozen one, counts as 1)
        f = np.maximum(first[order], 1)
        b = [int(np.searchsorted(f, c, side='left')) for c in cuts[:-1]] + [nt]
        b[0] = 0
    else:
        b = shard_bounds(L[order], world_size)
    return [int(k) for k in order[b[rank]:b[rank + 1]]]


def _stream_of(backend):
    """The stream context of a HipBackend (its kernels and the collectives on its buffers share one
    stream); nothing for host-side test doubles."""
    ctx = getattr(backend, 'stream_ctx', None)
    return ctx() if ctx is not None else contextlib.nullcontext()


TRIAL_TIMED_OUT_WORD = float(1 << 40)      # kTrialTimedOutWord of csrc/ba_types.h


def trial_status_of_sum(total, world, own_parts):
    """Solver status from the status word of the shards' trial records after their SUM over the ranks (csrc/ba_types.h
    trial_status_of_sum): a rank whose solve timed out wrote 2^40 instead of its status, more than any sum of pivot indices,
    so a time-out stays a time-out (SOLVE_TIMED_OUT) instead of turning into a pivot index.  Otherwise every rank solved the
    same system (total / world) or, with the solve spread over the ranks (own_parts), its own part: any non-zero = failed."""
    from ._capi import SOLVE_TIMED_OUT
    if total >= TRIAL_TIMED_OUT_WORD:
        return SOLVE_TIMED_OUT
    if total == 0:
        return 0
    if own_parts:
        return int(round(max(1., min(2e9, abs(total)))))
    return int(round(total / world))


class ShardComm(object):
    """The collectives the sharded adjuster needs, over torch.distributed
    ('nccl' = RCCL on ROCm for GPU tensors, 'gloo' for the CPU tests)."""

    def __init__(self, group=None, device=None, collectives='library'):
        """collectives: 'library' = the data-path collectives are issued by libpysfm_ba itself (RCCL on the
        handle's own stream, ba_comm_*) whenever that is possible; 'torch' = always through torch.distributed."""
        assert collectives in ('library', 'torch')
        self.collectives = collectives
        self.direct_fallback_reason = None
        import torch
        import torch.distributed as dist
        assert dist.is_initialized(), 'call torch.distributed.init_process_group first'
        self._torch, self._dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)
        self.local_rank = int(os.environ.get('LOCAL_RANK', self.rank))
        backend = dist.get_backend(group)
        if device is None:
            device = torch.device('cuda', self.local_rank) if backend == 'nccl' else torch.device('cpu')
        self.device = device
        self.bytes_reduced = 0

    def enable_direct(self, backend):
        """Move the data-path collectives into the library (ba_comm_*: RCCL issued on the handle's own stream,
        a sharded trial is then ONE C call).  Collective: every rank calls it, and every decision on the way is
        agreed on over torch.distributed, so that either all ranks switch or none does.
        ShardComm(collectives='torch') keeps the torch.distributed path.  Returns True when the library took
        over; otherwise `direct_fallback_reason` says why not, and anything other than a deliberate choice is
        reported on stderr (a broken direct path must not hide behind a merely slower number)."""
        torch, dist = self._torch, self._dist
        self.direct = None
        why = None
        if self.collectives == 'torch':
            why = "collectives='torch' requested"
        elif dist.get_backend(self.group) != 'nccl':
            why = 'process group backend is %s, not nccl (RCCL)' % dist.get_backend(self.group)
        elif not hasattr(backend, 'comm_attach'):
            why = 'backend %s has no collectives of its own' % type(backend).__name__
        elif not backend.comm_load(os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')):
            why = 'ba_comm_load could not resolve RCCL from torch\'s librccl.so'
        if not self._agree(why is None):
            return self._stay_on_torch(why or 'another rank could not switch', quiet=why is not None and (
                self.collectives == 'torch' or dist.get_backend(self.group) != 'nccl' or not hasattr(backend, 'comm_attach')))
        ident = torch.zeros(128, dtype=torch.uint8, device=self.device)
        if self.rank == 0:
            ident.copy_(torch.frombuffer(bytearray(backend.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(ident, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        try:
            backend.comm_attach(bytes(ident.cpu().numpy().tobytes()), self.rank, self.world_size)
            got = backend.comm_allreduce_sum([self.rank + 1.0, 1.0])           # self-test against the known answer
            ok = abs(got[0] - self.world_size * (self.world_size + 1) / 2.0) < 1e-9 and abs(got[1] - self.world_size) < 1e-9
            if not ok:
                why = 'self-test all-reduce returned %r' % (list(got),)
        except Exception as e:                                                   # noqa: BLE001 - any failure: stay on torch, loudly
            ok = False
            why = 'ba_comm_init / self-test failed: %s: %s' % (type(e).__name__, e)
        if not self._agree(ok):
            if getattr(backend, 'direct_comm', False):
                backend.comm_detach()
            return self._stay_on_torch(why or 'another rank failed its self-test', quiet=False)
        self.direct = backend
        return True

    def _stay_on_torch(self, why, quiet):
        self.direct_fallback_reason = why
        if not quiet:
            sys.stderr.write('[pysfm_amd rank %d] collectives stay on torch.distributed: %s\n' % (self.rank, why))
        return False

    def _agree(self, flag):
        t = self._torch.tensor([1.0 if flag else 0.0], dtype=self._torch.float64, device=self.device)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MIN, group=self.group)
        return bool(t.item() > 0.5)

    def allreduce_scalar(self, x):
        t = self._torch.tensor([x], dtype=self._torch.float64, device=self.device)
        self._dist.all_reduce(t, group=self.group)
        return float(t.item())

    def allreduce_max(self, x):
        t = self._torch.tensor([float(x)], dtype=self._torch.float64, device=self.device)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def allreduce_array(self, a):
        t = self._torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).to(self.device)
        self._dist.all_reduce(t, group=self.group)
        return t.cpu().numpy().reshape(np.shape(a))

    def allreduce_reduced(self, backend):
        """The one data-path collective: sum the partial reduced camera systems."""
        if getattr(self, 'direct', None) is backend:
            backend.comm_allreduce_reduced()        # RCCL inside the library, on its own stream
            return
        payload = backend.reduced_payload()
        with _stream_of(backend):              # ordered after the kernels that wrote it, before the ones that read it
            self._all_reduce_device(payload)
        self.bytes_reduced += payload.numel() * 8

    def allreduce_exchange(self, backend, t):
        """One of the three small sums of the distributed reduced solve (HipBackend.dist_stage), in place."""
        with _stream_of(backend):
            self._all_reduce_device(t)
        self.bytes_reduced += t.numel() * 8

    def allreduce_trial_result(self, backend, npartials):
        """Sum of the ranks' trial costs from the device buffer HipBackend.trial_result()
        (no host round trip before the collective); ONE synchronisation for the three numbers.
        Returns (cost, singular point blocks over all ranks, solver status)."""
        torch = self._torch
        with _stream_of(backend):
            # the whole record (cost partials | singular blocks | solver status) is summed over the ranks in
            # place - 16 KB, the same latency as 8 bytes - and read back with ONE copy; the partials are added
            # on the host in index order.  Every rank solves the same reduced system: status / world = status.
            t = backend.trial_result()
            self._all_reduce_device(t)
            if getattr(self, '_trial_host', None) is None or self._trial_host.numel() != t.numel():
                self._trial_host = torch.empty(t.numel(), dtype=t.dtype).pin_memory() if t.is_cuda else torch.empty_like(t)
            self._trial_host.copy_(t, non_blocking=True)
            if t.is_cuda:
                torch.cuda.current_stream().synchronize()
        h = self._trial_host.numpy()
        world = self._dist.get_world_size(self.group)
        return float(h[:npartials].sum()), int(round(h[npartials])), trial_status_of_sum(h[npartials + 1], world, getattr(backend, 'dist_on', False))

    def _all_reduce_device(self, t):
        """Sum a device tensor over the ranks in place.  RCCL does it on the device; a gloo group (the
        two-ranks-on-one-GPU test) stages it through the host."""
        if t.is_cuda and self._dist.get_backend(self.group) != 'nccl':
            h = t.cpu()
            self._dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            self._dist.all_reduce(t, group=self.group)

    def barrier(self):
        self._dist.barrier(group=self.group)

    def gather_points(self, adjuster):
        """Assemble the full reconstruction on every rank (each rank updates only its
        own tracks).  Returns reconstruction[nt_total, 3]."""
        b = adjuster.bundle
        full = np.zeros_like(np.asarray(b.reconstruction, float))
        ids = np.asarray(adjuster.track_ids, int)
        full[ids] = np.asarray(b.reconstruction, float)[ids]
        return self.allreduce_array(full)
