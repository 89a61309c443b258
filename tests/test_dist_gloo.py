"""World-size-2 test of the sharded adjuster on CPU (gloo): tracks split across two
processes, the reduced camera system summed with one all-reduce per trial, the LM
trajectory identical to the unsharded run.  The arithmetic comes from the
OracleBackend test double; what is under test is pysfm_amd.distributed + the comm hooks
of BundleAdjuster (the same code path the RCCL run uses)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle_backend import OracleBackend
    from pysfm_amd import Bundle, BundleAdjuster, sensor_model
    from pysfm_amd import synthetic_data as sd
    from pysfm_amd.distributed import ShardComm, shard_tracks
    s = sd.generate_banded_scene(14, 90, track_len=5, outlier_frac=.05)
    b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'],
                                sensor_model=sensor_model.CauchyModel(.05))
    comm = ShardComm()
    ids = shard_tracks(b, rank, world)
    ba = BundleAdjuster(backend=OracleBackend(), comm=comm, verbose=False)
    ba.set_bundle(b, track_ids=ids)
    cost0 = ba.compute_cost(b)
    ba.prepare_schur_complement()
    ba.apply_damping(3.)
    S, rhs = ba.compute_schur_complement()
    HCC = ba.HCCs
    ba.optimize(max_steps=6)
    X = comm.gather_points(ba)
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), ids=np.array(ids), cost0=cost0, S=S, b=rhs, HCC=HCC,
             costs=np.array(ba.costs), R=ba.bundle.Rs(), t=ba.bundle.ts(), X=X, nbytes=comm.bytes_reduced,
             trials=ba.lm_trials)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_adjuster_equals_single(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = dict(np.load(tmp_path / 'rank0.npz'))
    r1 = dict(np.load(tmp_path / 'rank1.npz'))
    # the shards are a partition into two contiguous ranges
    assert r0['ids'][0] == 0 and r1['ids'][-1] == 89 and r0['ids'][-1] + 1 == r1['ids'][0]
    assert abs(len(r0['ids']) - len(r1['ids'])) <= 2

    sys.path.insert(0, HERE)
    from oracle_backend import OracleBackend
    from pysfm_amd import Bundle, BundleAdjuster, sensor_model
    from pysfm_amd import synthetic_data as sd
    s = sd.generate_banded_scene(14, 90, track_len=5, outlier_frac=.05)
    b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'],
                                sensor_model=sensor_model.CauchyModel(.05))
    ba = BundleAdjuster(b, backend=OracleBackend(), verbose=False)
    cost0 = ba.compute_cost(b)
    ba.prepare_schur_complement()
    ba.apply_damping(3.)
    S, rhs = ba.compute_schur_complement()
    HCC = ba.HCCs
    ba.optimize(max_steps=6)

    for r in (r0, r1):                                  # every rank holds the full reduced system
        assert abs(r['cost0'] - cost0) <= 1e-12 * cost0
        assert np.max(np.abs(r['S'] - S)) <= 1e-12 * np.max(np.abs(S))
        assert np.max(np.abs(r['b'] - rhs)) <= 1e-12 * np.max(np.abs(rhs))
        assert np.max(np.abs(r['HCC'] - HCC)) <= 1e-12 * np.max(np.abs(HCC))
        assert len(r['costs']) == len(ba.costs)
        assert np.allclose(r['costs'], ba.costs, rtol=1e-9, atol=0)
        assert np.allclose(r['R'], ba.bundle.Rs(), rtol=0, atol=1e-9)
        assert np.allclose(r['t'], ba.bundle.ts(), rtol=0, atol=1e-9)
        assert np.allclose(r['X'], ba.bundle.reconstruction, rtol=0, atol=1e-9)
        # exactly one payload all-reduce per LM trial (+1 for the manual compute_schur_complement)
        nco = 13
        assert r['nbytes'] == (int(r['trials']) + 1) * (nco * nco * 36 + nco * 6) * 8
    assert np.array_equal(r0['costs'], r1['costs'])      # replicated decisions are bit-identical


def test_shard_bounds_balance_observations():
    from pysfm_amd.distributed import shard_bounds
    L = np.array([10] * 50 + [1] * 500)
    b = shard_bounds(L, 4)
    assert b[0] == 0 and b[-1] == 550 and all(x <= y for x, y in zip(b, b[1:]))
    sums = [L[b[i]:b[i + 1]].sum() for i in range(4)]
    assert max(sums) - min(sums) <= 10
    assert shard_bounds([3, 3], 1) == [0, 2]
    assert shard_bounds(np.zeros(0, int), 2) == [0, 0, 0]
