#!/usr/bin/env python3
"""bench.py - LM-iteration throughput of the MI355X bundle-adjustment inner loop.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the configuration the metric is quoted on):
1000 cameras / 100 000 points / 1 000 000 observations, full Levenberg-Marquardt
(lambda0 = 10, x0.1 / x10), Gaussian sensor model, synthetic banded scene
(pysfm_amd.synthetic_data.generate_banded_scene, seed 654), camera 0 frozen.
With N > 1 GPUs the points are sharded (weak scaling: 100 000 points / 1M
observations PER GPU, the 1000 cameras replicated) and the reduced camera system is
summed with one RCCL all-reduce per trial.

One "step" = one complete LM trial, nothing cached or skipped:
  linearise (residuals + 2x6/2x3 Jacobians + block assembly)  -> damp -> per-point 3x3
  pinv -> Schur reduction -> [all-reduce] -> reduced solve -> back-substitution ->
  parameter update on the trial set -> trial cost -> accept / reject.
value = observations x steps / wall time, summed over all GPUs (max over ranks of the
time).  Inputs are resident in HBM before the timed region starts.

Rank 0 prints ONE JSON line; see DESIGN.md for `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
FP64_MATRIX_PEAK_TFLOPS = 78.6  # MI355X dense fp64 matrix (= vector) peak; v_mfma_f64_16x16x4_f64 measured at 16 FMA/clk/SIMD


def schur_flops(nobs, nt):
    """Useful fp64 flops of one Schur reduction: per point with L observations, L products
    T = W HPPinv (6x3x3) and L(L+1)/2 block products T W^T (6x3x6), 2 flops per FMA."""
    L = nobs / max(nt, 1)
    return nt * (L * 54 + L * (L + 1) / 2 * 108) * 2


def algorithmic_bytes(kernel, nc, nco, nt, nobs, nunits, hb):
    """HBM bytes one launch of `kernel` has to move, every array touched once
    (fp64 values, int32 indices, our SoA layout, block-band S)."""
    obs = 20 * nobs                       # obs_cam (4) + obs_z (16); obs_pt only in k_cost
    cams, pts = 96 * nc, 24 * nt
    band = 288 * nco * (hb + 1)           # reduced system S in block-band layout
    if kernel == 'schur_pairs':           # S and b are read-modify-written once
        return obs + 12 * nunits + 4 * nt + cams + pts + 72 * nt + 2 * band + 2 * 48 * nco
    if kernel == 'linearize':             # point blocks: HPP (48) + bP (24) written per track
        return obs + 4 * nt + cams + pts + 72 * nt
    if kernel == 'camera_blocks':         # camera-ordered pass: perm (4) + obs_pt (4) + obs_z (16) per obs
        return 24 * nobs + cams + pts + 2 * 336 * nc
    if kernel == 'cost':
        return 24 * nobs + cams + pts + 4 * nc + nt
    if kernel == 'backsub':
        return obs + 4 * nt + cams + pts + 72 * nt + 48 * nco + 24 * nt
    if kernel == 'schur_init':
        return band + 336 * nc + 288 * nco
    if kernel == 'band_solve':            # read S, write U, re-read U (backward pass)
        return 3 * band + 4 * 48 * nco
    if kernel in ('bcr_assemble', 'bcr_eliminate', 'bcr_backsolve'):
        # block cyclic reduction over N super-blocks of B = 6 hb unknowns; per AVERAGE launch
        # (levels = ceil(log2 N) launches share the N nodes)
        B = 6 * max(hb, 1)
        N = -(-nco // max(hb, 1))
        levels = max(1, int(np.ceil(np.log2(max(N, 2)))))
        if kernel == 'bcr_assemble':
            return band + 8 * (2 * N * B * B + N * B)
        if kernel == 'bcr_eliminate':     # read D, T[l,i], T[i,r]; write G^-1, P, Q, T[l,r]; RMW D_l, D_r
            return 8 * N * (11 * B * B + 6 * B) // levels
        return 8 * N * (3 * B * B + 4 * B) // levels
    if kernel == 'flatten':
        return band + 288 * nco * nco
    if kernel == 'point_invert':
        return 96 * nt
    if kernel == 'update':
        return 2 * (96 * nc + 24 * nt) + 48 * nco + 24 * nt
    return 0


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary
    (profiles/*_hbm_traffic.csv, written by scripts/summarize_profile.py from separate
    --pmc FETCH_SIZE / WRITE_SIZE passes of this same command).  None if absent."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_hbm_traffic.csv')))
    if not files:
        return None, None
    # the timer id 'schur_pairs' covers the interchangeable reduction kernels
    names = ['k_schur_groups_mfma2', 'k_schur_groups_mfma', 'k_schur_groups', 'k_schur_pairs'] if kernel == 'schur_pairs' else ['k_' + kernel]
    rows = list(csv.DictReader(open(files[-1])))
    for name in names:
        for row in rows:
            if row['kernel'].split('<')[0].endswith(name):
                return int(row[[k for k in row if k.startswith('hbm_bytes')][0]]), os.path.basename(files[-1])
    return None, None


def cpu_baseline(sample_cams, sample_pts):
    """The oracle (NumPy restatement of the reference, 'port') timed on this box's host
    cores on a bounded sample of the same workload: one full LM trial."""
    from oracle import ba_oracle as O
    from pysfm_amd import synthetic_data as sd
    s = sd.generate_banded_scene(sample_cams, sample_pts)
    flags = (np.arange(sample_cams, dtype=np.int32) - 1, np.ones(sample_pts, bool))
    a = (s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
    sen = O.Sensor.gaussian(1.)
    t0 = time.time()
    c0 = O.cost(sen, *a, *flags)
    mu, su = O.compute_update(sen, *a, *flags, damping=10.)
    R2, t2, X2 = O.apply_update(s['R0'], s['t0'], s['X0'], mu, su, *flags)
    c1 = O.cost(sen, s['K'], R2, t2, X2, *a[4:], *flags)
    dt = time.time() - t0
    try:
        import threadpoolctl
        threads = max([p.get('num_threads', 1) for p in threadpoolctl.threadpool_info()] + [1])
    except Exception:
        threads = 1
    return dict(value=len(s['obs_cam']) / dt, unit='obs/s', cores=int(threads), kind='port',
                sample='one full LM trial of oracle/ba_oracle.py (NumPy) on a %d-camera / %d-point / %d-observation '
                       'scene from the same generator, %.1f s; host has %d cores, BLAS threads=%d; '
                       'cost %.4f -> %.4f' % (sample_cams, sample_pts, len(s['obs_cam']), dt, os.cpu_count(), threads, c0, c1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--cams', type=int, default=1000)
    ap.add_argument('--pts-per-gpu', type=int, default=100000)
    ap.add_argument('--sensor', default='gaussian', choices=['gaussian', 'cauchy', 'huber'])
    ap.add_argument('--outliers', type=float, default=0.)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-table', action='store_true',
                    help='no HIP events at all in the timed region (for external kernel traces); roofline uses the warm-up timings')
    ap.add_argument('--no-lm', action='store_true', help='skip the untimed full optimize() that yields the RMSE')
    args = ap.parse_args()

    import torch
    from pysfm_amd import Bundle, BundleAdjuster, sensor_model
    from pysfm_amd import synthetic_data as sd
    from pysfm_amd._capi import PARAMS_CUR

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    comm = None
    if world > 1 or os.environ.get('BA_FORCE_COMM'):      # BA_FORCE_COMM: exercise the RCCL path on one GPU
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        from pysfm_amd.distributed import ShardComm, shard_tracks
        comm = ShardComm()
    ngpus = world
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (no CPU path)'

    # ---- scene (identical on every rank), then this rank's shard of the tracks
    nc, nt = args.cams, args.pts_per_gpu * ngpus
    s = sd.generate_banded_scene(nc, nt, outlier_frac=args.outliers)
    model = {'gaussian': sensor_model.GaussianModel(1.), 'cauchy': sensor_model.CauchyModel(.05),
             'huber': sensor_model.HuberModel(.06)}[args.sensor]
    bundle = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'],
                                     sensor_model=model)
    ba = BundleAdjuster(device=local_rank, comm=comm, verbose=False)
    track_ids = None
    if comm is not None:
        track_ids = shard_tracks(bundle, rank, world)
    ba.set_bundle(bundle, track_ids=track_ids)
    be = ba.backend
    nobs_local = be.nobs
    nobs_total = len(s['obs_cam'])

    def sync():
        torch.cuda.synchronize()
        if comm is not None:
            comm.barrier()
            torch.cuda.synchronize()

    # ---- (untimed) the full LM run of config 3: final cost + reprojection RMSE
    lm = {}
    if not args.no_lm:
        sync()
        t0 = time.time()
        ba.optimize(max_steps=25)
        sync()
        lm_wall = time.time() - t0
        e = be.eval_observations(PARAMS_CUR, e=True, r=False, Jc=False, Jp=False)['e']
        sq, cnt = float(np.sum(e * e)), float(len(e))
        if comm is not None:
            sq, cnt = comm.allreduce_scalar(sq), comm.allreduce_scalar(cnt)
        lm = dict(final_reproj_rmse=float(np.sqrt(sq / cnt)), lm_steps=ba.num_steps,
                  lm_trials=int(ba.lm_trials), lm_converged=bool(ba.converged),
                  lm_cost_initial=ba.costs[0], lm_cost_final=ba.costs[-1], lm_wall_s=lm_wall)
        # restart from the initial guess for the timed trials
        ba.set_bundle(bundle, track_ids=track_ids)
        be = ba.backend

    # ---- timed region: K complete LM trials, continuing the LM schedule
    state = dict(damping=10., cur=None, paths={})

    def one_trial():
        if state['cur'] is None:
            state['cur'] = ba._cost(PARAMS_CUR)
        accepted, nxt = ba.trial(state['damping'], None, state['cur'])
        key = '%s/%s' % (getattr(be, 'last_solve_path', '?'), 'accepted' if accepted else ('rejected' if accepted is not None else 'ill-conditioned'))
        state['paths'][key] = state['paths'].get(key, 0) + 1
        if accepted:
            state['damping'] *= .1
            state['cur'] = nxt
        else:
            state['damping'] *= 10.
        if state['damping'] >= 1e8 or state['damping'] < 1e-12:      # schedule exhausted: restart it
            state['damping'] = 10.

    # warm-up with every kernel bracketed by HIP events: finds the dominant kernel
    be.enable_timing(True)
    be.timings(reset=True)
    nwarm = 0
    for i in range(args.warmup):
        one_trial()
        nwarm += 1
        if i == 0 and args.warmup > 1:          # the very first launches include lazy code-object loading
            be.timings(reset=True)
            nwarm = 0
    tm_w = be.timings(reset=True)
    cand = {k: v for k, v in tm_w.items() if v['launches'] > 0}
    dom = max(cand, key=lambda k: cand[k]['ms']) if cand else 'schur_pairs'
    # timed region: only the dominant kernel keeps its events (an event pair costs a few
    # microseconds of stream time - bracketing all ~25 launches of a 0.6 ms step would slow it ~15 %)
    be.enable_timing(not args.no_kernel_table, only=[dom], stride=4)     # every 4th step: the events themselves cost stream time
    sync()
    state['paths'] = {}
    t0 = time.time()
    for _ in range(args.steps):
        one_trial()
    sync()
    dt = time.time() - t0
    timed_paths = dict(state['paths'])
    tm_dom = be.timings(reset=True)[dom]
    if args.no_kernel_table:
        tm_dom, tm, nprof = tm_w[dom], tm_w, max(1, nwarm)
    else:
        # after the timed region: a few more trials with everything bracketed, for the per-kernel table
        be.enable_timing(True)
        nprof = max(3, min(10, args.steps))
        for _ in range(nprof):
            one_trial()
        tm = be.timings(reset=True)
    be.enable_timing(False)
    if comm is not None:
        dt = comm_max(comm, dt)
    ms_per_step = 1e3 * dt / args.steps
    value = nobs_total * args.steps / dt

    out = None
    if rank == 0:
        # dominant kernel of OUR kernels, by HIP-event time on the launch stream
        ours = {k: v for k, v in tm.items() if v['launches'] > 0}
        # the kernel the timed region bracketed was picked from the warm-up; the table measured after the timed
        # region (every kernel bracketed, `nprof` trials) has the last word on which kernel dominates
        dom_table = max(ours, key=lambda k: ours[k]['ms']) if ours else dom
        if dom_table != dom or not tm_dom['launches'] or not tm_dom['ms'] > 0.:
            dom, tm_dom = dom_table, ours.get(dom_table, tm_dom)
        avg_ms = tm_dom['ms'] / max(1, tm_dom['launches'])
        nco = be.nco
        B = algorithmic_bytes(dom, be.nc, nco, be.nt, nobs_local, be.nt, be.half_bandwidth)
        achieved = B / (avg_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(dom)
        copy_gbs = be.measure_copy_bandwidth()
        out = {
            'metric': 'LM-iter throughput (obs/sec) + final reproj RMSE, 1k-cam/100k-pt/1M-obs scene',
            'value': value, 'unit': 'obs/s', 'n_gpus': ngpus, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[2]: %d cameras / %d points / %d observations%s, full LM trial per step '
                                   '(linearise+damp+pinv+Schur%s+reduced solve+backsub+update+cost), %s sensor model, camera 0 frozen'
                                   % (nc, nt, nobs_total, ' (%d points / %d obs per GPU)' % (args.pts_per_gpu, nobs_local) if ngpus > 1 else '',
                                      '+RCCL all-reduce' if ngpus > 1 else '', args.sensor),
                       'cameras': nc, 'points': nt, 'observations': nobs_total, 'parallelism': 'points sharded x%d' % ngpus,
                       'collectives': None if comm is None else ('RCCL inside the library (ba_comm_*)' if getattr(be, 'direct_comm', False) else 'torch.distributed (RCCL)')},
            'roofline': {'bound': 'hbm', 'kernel': 'k_' + dom, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_src,
                         'measured_copy_GBps': copy_gbs, 'frac_of_measured_copy': achieved / copy_gbs,
                         'algorithmic_bytes_per_launch': B,
                         'avg_launch_ms': avg_ms, 'launches': tm_dom['launches'],
                         'note': 'HIP events on the launch stream during the timed steps, every 4th step; back-to-back launches of one kernel (the cyclic-reduction levels) share one event pair, avg = elapsed / launches'},
            'matrix_cores': {'kernel': 'k_schur_groups_mfma2 (timer schur_pairs)', 'useful_flops_per_launch': schur_flops(nobs_local, be.nt),
                             'achieved_tflops': schur_flops(nobs_local, be.nt) / max(1e-9, ours['schur_pairs']['ms'] / max(1, ours['schur_pairs']['launches']) * 1e-3) / 1e12
                             if 'schur_pairs' in ours else None,
                             'peak_tflops': FP64_MATRIX_PEAK_TFLOPS,
                             'note': 'fp64 MFMA is used where the path is GEMM-shaped (Schur reduction, cyclic-reduction nodes); informational'},
            'kernel_ms_per_step': {k: v['ms'] / nprof for k, v in ours.items()},
            'kernel_launches_per_step': {k: v['launches'] / nprof for k, v in ours.items()},
            'all_kernels': {'algorithmic_bytes_per_step': int(sum(
                                algorithmic_bytes(k, be.nc, nco, be.nt, nobs_local, be.nt, be.half_bandwidth) * v['launches']
                                for k, v in ours.items()) / nprof),
                            'kernel_ms_per_step': sum(v['ms'] for v in ours.values()) / nprof,
                            'note': 'measured on %d extra trials outside the timed region, every kernel bracketed' % nprof},
            'reduced_system': {'cameras_optimised': nco, 'block_half_bandwidth': be.half_bandwidth,
                               'bytes': 8 * be.S_doubles, 'solve_path': getattr(be, 'last_solve_path', None),
                               'timed_trials_by_solver_and_outcome': timed_paths},
        }
        out.update(lm)
        if ngpus == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(300, 30000)
    if comm is not None:
        import ctypes
        import torch.distributed as dist
        ctypes.CDLL(None).fflush(None)          # (every rank's buffered C output is out before rank 0 prints)
        dist.barrier()
        dist.destroy_process_group()
    # RCCL writes a version banner to the C stdout buffer, which is flushed at exit, i.e. AFTER anything Python
    # prints: flush it now so that the JSON line is the last line of stdout
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if rank == 0:
        print(json.dumps(out), flush=True)


def comm_max(comm, x):
    import torch
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device=comm.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


if __name__ == '__main__':
    main()
