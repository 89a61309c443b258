#!/usr/bin/env python3
"""bench.py - LM-iteration throughput of the MI355X bundle-adjustment inner loop.

    python bench.py --gpus 1 --steps 20 --warmup 20
    python bench.py --gpus 8                       # starts its own 8 ranks (torch.distributed.run, 127.0.0.1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workloads (BASELINE.json `configs`; synthetic banded scene of pysfm_amd.synthetic_data, seed 654, camera 0 frozen):
  --config 3  (default, the configuration the metric is quoted on) 1000 cameras / 100 000 points / 1 000 000
              observations, full Levenberg-Marquardt (lambda0 = 10, x0.1 / x10), Gaussian sensor model.
              With N > 1 GPUs: WEAK scaling - 100 000 points / 1M observations PER GPU, the 1000 cameras replicated.
  --config 4  config 3 with the Huber robustifier (k = 0.06) and 10 % gross outliers.
  --config 2  100 cameras / 10 000 points / 100 000 observations.
  --config 5  10 000 cameras / 1 000 000 points / 10 000 000 observations in total, STRONG split of the points over
              the N GPUs (N = 1 runs the whole scene on one GPU).
  --shuffle-points   hand the tracks (and their observations) over in random order: the library orders them itself.
  --track-len L      observations per point (default 10; the band half-width of the reduced system is L - 1).
Points are sharded over the GPUs; the reduced camera system is summed with one RCCL all-reduce per trial.

One "step" = one complete LM trial, nothing cached or skipped:
  linearise (residuals + 2x6/2x3 Jacobians + block assembly)  -> damp -> per-point 3x3
  pinv -> Schur reduction -> [all-reduce] -> reduced solve -> back-substitution ->
  parameter update on the trial set -> trial cost -> accept / reject.
value = observations x steps / wall time, summed over all GPUs (max over ranks of the
time).  Inputs are resident in HBM before the timed region starts.

Rank 0 prints ONE short JSON line (<= 4 KB: the contract's keys, `roofline`, `cpu_baseline`) as the last line of stdout and writes
everything else it measured to bench_detail.json (--detail-out); see DESIGN.md for `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
FP64_MATRIX_PEAK_TFLOPS = 78.6  # MI355X dense fp64 matrix (= vector) peak; v_mfma_f64_16x16x4_f64 measured at 16 FMA/clk/SIMD

# What binds each kernel at the bench sizes; `roofline.limited_by` reports it.  The contract prices
# `achieved` against HBM for every kernel that is not MFMA-bound, so `frac` is always achieved / 8 TB/s.
KERNEL_BOUND = {'border_schur': 'atomics', 'border_solve': 'latency', 'bcr_eliminate': 'latency', 'bcr_backsolve': 'latency', 'bcr_refine': 'latency', 'bcr_assemble': 'hbm', 'band_solve': 'latency',
                'dense_solve': 'latency', 'pcg_solve': 'hbm', 'schur_pairs': 'mfma', 'linearize': 'latency', 'backsub': 'latency', 'cost': 'hbm',
                'point_invert': 'hbm', 'schur_init': 'hbm', 'camera_blocks': 'hbm', 'update': 'hbm', 'flatten': 'hbm'}
# timer id (include/pysfm_ba.h BA_K_*) -> the kernels that run under it on the product path
KERNEL_NAMES = {'border_schur': 'k_schur_border (+ k_border_clear)', 'border_solve': 'k_border_prepare, k_bcr_apply (a launch per level, forward and back), k_border_reduce, k_border_solve, k_border_correct',
                'linearize': 'k_linearize_groups_trial (with the point inverses and the clearing of [S | b]; k_linearize_groups / k_linearize outside a trial or when points do not come in runs)', 'point_invert': 'k_point_invert_schur_init (the trials after a rejected one)',
                'schur_pairs': 'k_schur_groups_mfma2 | k_schur_groups_mfma3 (k_schur_groups / k_schur_pairs otherwise)',
                'backsub': 'k_backsub_groups (k_backsub when points do not come in runs)',
                'bcr_eliminate': 'k_bcr_eliminate_fused: all elimination levels of the cyclic reduction AND its back-substitution in one launch (k_bcr_eliminate per level where a level is wider than the chip; k_bcrw_* for half-bandwidths 12..23)',
                'bcr_backsolve': 'k_bcr_backsolve_fused (k_bcr_backsolve / k_bcrw_backsolve per level otherwise)',
                'bcr_refine': 'k_bcr_residual + k_bcr_refine: one step of iterative refinement through the kept factors (trials damped below 1e-2)', 'bcr_assemble': 'k_bcr_assemble', 'cost': 'k_cost',
                'camera_blocks': 'k_camera_blocks', 'dense_solve': 'k_dense_gather/panel/update/backsolve', 'band_solve': 'k_band_solve'}
# reference rates measured in SURVEY.md section 6 (the reference itself, imported in the build container, 1 core Xeon 2.1 GHz)
SURVEY_REFERENCE_RATES = {'assemble_obs_per_s': 2.8e4, 'whole_update_obs_per_s': 3.5e3,
                          'source': 'SURVEY.md section 6: alexflint/pysfm bundle_adjuster.py on 1 core (Xeon 2.1 GHz)'}


def schur_flops(nobs, nt):
    """Useful fp64 flops of one Schur reduction: per point with L observations, L products
    T = W HPPinv (6x3x3) and L(L+1)/2 block products T W^T (6x3x6), 2 flops per FMA."""
    L = nobs / max(nt, 1)
    return nt * (L * 54 + L * (L + 1) / 2 * 108) * 2


FP64_RIDGE_FLOP_PER_BYTE = FP64_MATRIX_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)      # 9.8: above it the fp64 pipe is the roof, not HBM
LINEARISE_FLOPS_PER_OBS = 450.   # projection, chain rule, Jc^T Jc, Jp^T Jp, Jc^T Jp, J^T r


BORDER_CAMS = [0]      # border cameras of the benched problem (problem_info): the border kernels' flops and bytes depend on it


def border_shape(nc, nco, nt, nobs, hb):
    """(pairs of observations the border blocks are formed from, nodes N and node size B of the band's cyclic reduction, row length ld of C)."""
    k = BORDER_CAMS[0]
    L = nobs / max(nt, 1)
    pairs = k * (nobs / max(nc, 1)) * L
    n1 = max(1, nco - k)
    cb = max(hb, 1)
    return pairs, -(-n1 // cb), 6 * cb, -(-6 * k // 16) * 16


def useful_flops(kernel, nc, nco, nt, nobs, hb):
    """Useful fp64 flops of one trial's launches of `kernel` (2 per FMA; what the algorithm needs ONCE - a kernel that
    linearises an observation again, pads its tiles or factors a block redundantly is not credited for it).
    SURVEY 8(d): 0.45 kflop per observation for the linearisation, (108 + 216 L) per observation for the Schur products
    over all ordered camera pairs of a track - S is symmetric, so the reduction is credited with the upper half,
    schur_flops()."""
    L = nobs / max(nt, 1)
    if kernel == 'linearize':
        return LINEARISE_FLOPS_PER_OBS * nobs
    if kernel == 'schur_pairs':
        return schur_flops(nobs, nt)
    if kernel == 'point_invert':          # damping, 3x3 eigen-solve (Jacobi sweeps), pinv, L D L^T: ~400 per point
        return 400. * nt
    if kernel == 'backsub':               # Jc dC (24), Jp^T (.) (12) per observation, HPPinv (.) per point, Rodrigues per camera, the trial cost
        return (36. + 60.) * nobs + 18. * nt + 120. * nc
    if kernel == 'cost':
        return 60. * nobs
    if kernel == 'border_schur':          # two linearisations and one 6x3x3 + 6x3x6 product per pair of observations
        return border_shape(nc, nco, nt, nobs, hb)[0] * (2 * LINEARISE_FLOPS_PER_OBS + 2 * (54 + 108))
    if kernel == 'border_solve':          # the border's columns through the tree (forward 3 B^2 ld, backward 6 B^2 ld per node, MACs x 2), C^T Y, the border system
        pairs, N, B, ld = border_shape(nc, nco, nt, nobs, hb)
        k6 = 6 * BORDER_CAMS[0]
        return 9. * N * B * B * ld + 2. * (12 * max(hb, 1) * k6) * k6 * k6 / 6 + k6 ** 3 / 3.
    if kernel == 'bcr_refine':            # r = b - S x over the band (2 flops per entry of the full band), then two sweeps of three B x B matrix-vector products per node
        B = 6. * max(hb, 1)
        N = -(-nco // max(hb, 1))
        return 2. * 36. * nco * (2 * hb + 1) + N * 12. * B * B
    if kernel in ('bcr_eliminate', 'bcr_backsolve', 'bcr_assemble'):
        # block cyclic reduction, N nodes of B = 6 hb unknowns: per node Cholesky B^3/3, P and Q (two triangular solves with B
        # right-hand sides, B^3 each), G^-1 (B^3/3), P^T P and Q^T Q (symmetric, B^3 each), the two couplings of the next
        # level (2 B^3 each): 8.67 B^3; back-substitution 3 matrix-vector products (6 B^2)
        B = 6. * max(hb, 1)
        N = -(-nco // max(hb, 1))
        if kernel == 'bcr_eliminate':
            return N * (26. / 3. * B ** 3 + 6. * B * B)
        if kernel == 'bcr_backsolve':
            return N * 6. * B * B
        return 0.
    return 0.


def minimal_bytes(kernel, nc, nco, nt, nobs, nunits, hb):
    """What `kernel` MUST move (inputs in, results out), where that is less than algorithmic_bytes: the reduced solve reads the
    band and the right-hand side and writes the solution - its D, U, P, Q, G^-1 workspace is the algorithm's own."""
    if kernel in ('bcr_eliminate', 'band_solve', 'dense_solve'):
        return 288 * nco * (hb + 1) + 2 * 48 * nco
    return algorithmic_bytes(kernel, nc, nco, nt, nobs, nunits, hb)


def algorithmic_bytes(kernel, nc, nco, nt, nobs, nunits, hb, launches=1.):
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
        # block cyclic reduction over N super-blocks of B = 6 hb unknowns; per AVERAGE launch: the N nodes of a solve are
        # shared by its `launches` launches (one with k_bcr_eliminate_fused / k_bcr_backsolve_fused, one per level without)
        B = 6 * max(hb, 1)
        N = -(-nco // max(hb, 1))
        if kernel == 'bcr_assemble':
            return band + 8 * (2 * N * B * B + N * B)
        if kernel == 'bcr_eliminate':     # read D, T[l,i], T[i,r]; write G^-1, P, Q, T[l,r]; RMW D_l, D_r
            return int(8 * N * (11 * B * B + 6 * B) / max(1., launches))
        return int(8 * N * (3 * B * B + 4 * B) / max(1., launches))
    if kernel == 'bcr_refine':            # per AVERAGE launch of its two: the band, b and x once; G^-1, P, Q of every node once per sweep
        B = 6 * max(hb, 1)
        N = -(-nco // max(hb, 1))
        return int((band + 3 * 48 * nco + 2 * 8 * N * (3 * B * B + 4 * B)) / max(1., launches))
    if kernel == 'flatten':
        return band + 288 * nco * nco
    if kernel == 'point_invert':          # HPP read, HPPinv + its factorisation written, [S | b] initialised
        return 48 * nt + 48 * nt + 72 * nt + 24 * nt + band + 48 * nco
    if kernel == 'update':
        return 2 * (96 * nc + 24 * nt) + 48 * nco + 24 * nt
    if kernel == 'border_schur':          # per pair: two observations (20 B each) and the point's inverse; the blocks written once
        pairs, N, B, ld = border_shape(nc, nco, nt, nobs, hb)
        return int(pairs * (2 * 20 + 72) + 288 * BORDER_CAMS[0] * (2 * hb + 1 + BORDER_CAMS[0]))
    if kernel == 'border_solve':          # C copied into F, F read and written once per level it takes part in (~4 passes), the factors G^-1, P, Q read twice
        pairs, N, B, ld = border_shape(nc, nco, nt, nobs, hb)
        return int(8 * (N * B * ld * (2 + 4) + 2 * 3 * N * B * B))
    return 0


PMC_WORKLOAD = 'config3'           # set by main(): which committed profile the traffic numbers may come from


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary OF THIS WORKLOAD
    (profiles/<round>_hbm_traffic.csv for config 3, profiles/<round>_config5_hbm_traffic.csv for config 5; written by
    scripts/summarize_profile.py from separate --pmc FETCH_SIZE / WRITE_SIZE passes of this same command).
    None if absent (other workloads have no committed counter run)."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_hbm_traffic.csv')))
    if PMC_WORKLOAD == 'config3':
        files = [f for f in files if '_config' not in os.path.basename(f)]
    else:
        files = [f for f in files if ('_%s_' % PMC_WORKLOAD) in os.path.basename(f)]
    if not files:
        return None, None
    # the timer id 'schur_pairs' covers the interchangeable reduction kernels
    names = {'schur_pairs': ['k_schur_groups_mfma2', 'k_schur_groups_mfma3', 'k_schur_wide_mfma', 'k_schur_rect_mfma', 'k_schur_groups_mfma', 'k_schur_groups', 'k_schur_pairs'],
             'linearize': ['k_linearize_groups_trial', 'k_linearize_groups', 'k_linearize'], 'backsub': ['k_backsub_groups', 'k_backsub'],
             'point_invert': ['k_point_invert_schur_init', 'k_point_invert'],
             'bcr_eliminate': ['k_bcr_eliminate_split', 'k_bcr_eliminate']}.get(kernel, ['k_' + kernel])
    rows = list(csv.DictReader(open(files[-1])))
    for name in names:
        for row in rows:
            if row['kernel'].split('<')[0].endswith(name):
                return int(row[[k for k in row if k.startswith('hbm_bytes')][0]]), os.path.basename(files[-1])
    return None, None


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(s, loop_obs=10000):
    """the oracle (NumPy restatement of the reference, kind 'port') timed on this box's host cores in
    the same run: (i) the literal per-observation assembly loop (the shape of the reference's own hot loop,
    bundle_adjuster.py:211-234) on a `loop_obs`-observation subsample, ONE core; (ii) the vectorised port, one full LM
    trial (cost, blocks, Schur, LU solve, back-substitution, update, trial cost) on the FULL scene, one process, BLAS
    threads as the box configures them.  `value` is (ii)."""
    from oracle import ba_oracle as O
    nc, nt = len(s['R0']), len(s['X0'])
    try:
        import threadpoolctl
        threads = max([p.get('num_threads', 1) for p in threadpoolctl.threadpool_info()] + [1])
    except Exception:
        threads = 1
    sen = O.Sensor.gaussian(1.)
    # (i) per-observation loop: the first `loop_obs` observations (whole tracks), their cameras and points
    n = int(min(loop_obs, len(s['obs_cam'])))
    n = int(np.searchsorted(s['obs_pt'], s['obs_pt'][n - 1], side='right')) if n else 0
    t0 = time.time()
    O.normal_blocks_loop(sen, s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'][:n], s['obs_pt'][:n], s['obs_z'][:n], nc, nt)
    dt_loop = time.time() - t0
    # (ii) vectorised port, full scene, one full LM trial
    flags = (np.arange(nc, dtype=np.int32) - 1, np.ones(nt, bool))
    a = (s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
    t0 = time.time()
    c0 = O.cost(sen, *a, *flags)
    t1 = time.time()
    blocks = O.normal_blocks(sen, *a, nc, nt)
    t_asm = time.time() - t1
    del blocks
    mu, su = O.compute_update(sen, *a, *flags, damping=10.)
    R2, t2, X2 = O.apply_update(s['R0'], s['t0'], s['X0'], mu, su, *flags)
    c1 = O.cost(sen, s['K'], R2, t2, X2, *a[4:], *flags)
    dt = time.time() - t0 - t_asm                      # (the stand-alone assembly above is timed on its own, not twice)
    N = len(s['obs_cam'])
    # (iii) the reference's own scale, BASELINE configs[0]: the whole optimize() of the 5-camera x 50-track scene of small_problems()
    from pysfm_amd import synthetic_data as sd1
    s1 = sd1.generate_banded_scene(5, 50, track_len=5, init_perturbation=.03)
    a1 = (s1['K'], s1['R0'], s1['t0'], s1['X0'], s1['obs_cam'], s1['obs_pt'], s1['obs_z'])
    f1 = (np.arange(5, dtype=np.int32) - 1, np.ones(50, bool))
    t1 = time.time()
    r1 = O.lm_optimize(sen, *a1, *f1)
    dt1 = time.time() - t1
    config1 = {'optimize_s': dt1, 'costs': len(r1['costs']), 'cores': int(threads),
               'sample': 'oracle lm_optimize (vectorised port) of the 5-camera / 50-track / 250-observation scene; small_problems.config1_* is the same scene on the GPU'}
    return dict(value=N / dt, unit='obs/s', cores=int(threads), kind='port', config1=config1,
                sample='one full LM trial of oracle/ba_oracle.py (vectorised NumPy port of the reference) on the FULL %d-camera / '
                       '%d-point / %d-observation scene: %.1f s (cost %.4f -> %.4f); assembly alone %.2f s'
                       % (nc, nt, N, dt, c0, c1, t_asm),
                per_observation_loop={'value': n / dt_loop, 'unit': 'obs/s', 'cores': 1,
                                      'sample': 'normal_blocks_loop (literal bundle_adjuster.py:211-234 loop) on the first %d observations, %.2f s' % (n, dt_loop)},
                vectorised_assembly_obs_per_s=N / t_asm,
                host={'cpu_model': cpu_model(), 'os_cpu_count': os.cpu_count(), 'blas_threads': int(threads)},
                reference_measured_in_survey=SURVEY_REFERENCE_RATES)


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n, one_gpu=False):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, one per GPU (one_gpu: all of them on GPU 0,
    --ranks-on-one-gpu)."""
    import torch
    have = torch.cuda.device_count()
    if have < n and not one_gpu:
        sys.stderr.write('bench.py: --gpus %d requested but only %d GPU(s) visible; refusing to run fewer ranks than asked for\n' % (n, have))
        sys.exit(2)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    sys.exit(subprocess.call(cmd, env=env))


def make_trial_runner(ba, be):
    """One LM trial per call, continuing the LM schedule (damping x0.1 on acceptance, x10 on rejection; restarted when exhausted)."""
    from pysfm_amd._capi import PARAMS_CUR
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
    return one_trial, state


def scene_variant(s, track_len, shuffle, drop):
    """The generator's scene with a fraction of its observations dropped at random (every track keeps two) and / or
    its tracks renumbered and its observations permuted at random."""
    obs_cam, obs_pt, obs_z, X0 = s['obs_cam'], s['obs_pt'], s['obs_z'], s['X0']
    nt = len(X0)
    if drop > 0:                                   # ragged tracks: points no longer share their camera lists
        rs = np.random.RandomState(11)
        keep = rs.rand(len(obs_cam)) >= drop
        keep[::track_len] = True
        keep[1::track_len] = True
        obs_cam, obs_pt, obs_z = obs_cam[keep], obs_pt[keep], obs_z[keep]
    if shuffle:                                    # tracks renumbered at random, observations in random order
        rs = np.random.RandomState(7)
        new_id = rs.permutation(nt)                # track k becomes track new_id[k]
        X0 = np.empty_like(s['X0'])
        X0[new_id] = s['X0']
        o = rs.permutation(len(obs_cam))
        obs_cam, obs_pt, obs_z = obs_cam[o], new_id[obs_pt[o]].astype(np.int32), obs_z[o]
    return obs_cam, obs_pt, obs_z, X0


def with_long_tracks(s, nc, nt, L, every, llen, seed=3):
    """The generator's scene with every `every`-th point seen by `llen` consecutive cameras instead of L (measurements from
    the true parameters + the scene's noise level)."""
    rs = np.random.RandomState(seed)
    long_pts = np.arange(every // 2, nt, every)
    keep = np.ones(len(s['obs_cam']), bool)
    keep.reshape(nt, L)[long_pts] = False
    c0 = np.clip(s['obs_cam'][long_pts * L] - llen // 2, 0, nc - llen)
    cam = (c0[:, None] + np.arange(llen)[None, :]).reshape(-1)
    pt = np.repeat(long_pts, llen)
    p = np.einsum('nij,nj->ni', s['R'][cam], s['X'][pt]) + s['t'][cam]
    z = p[:, :2] / p[:, 2:3] + rs.randn(len(cam), 2) * .02
    oc = np.concatenate((s['obs_cam'][keep], cam)).astype(np.int32)
    op = np.concatenate((s['obs_pt'][keep], pt)).astype(np.int32)
    oz = np.concatenate((s['obs_z'][keep], z))
    o = np.lexsort((oc, op))
    out = dict(s)
    out.update(obs_cam=oc[o], obs_pt=op[o], obs_z=oz[o])
    return out


def with_cameras_renumbered(s, seed=5):
    """The same scene with its cameras renumbered at random (ids as a database assigns them, an unordered image collection):
    camera c becomes camera perm[c].  The caller's order of the optimised cameras then spreads every track over most of the
    sequence; ba_set_problem orders the cameras itself (pysfm_amd/csrc/ba_order.hip)."""
    rs = np.random.RandomState(seed)
    nc = len(s['R0'])
    perm = rs.permutation(nc)
    out = dict(s)
    for k in ('R0', 't0', 'R', 't'):
        a = np.empty_like(s[k])
        a[perm] = s[k]
        out[k] = a
    out['obs_cam'] = perm[s['obs_cam']].astype(np.int32)
    return out


def with_loop_closures(s, n=10, seed=21):
    """The scene with n extra tracks that each tie a camera i to the camera half a sequence further on (a loop closure): the
    band of the reduced system would become as wide as that; the library moves the far cameras to a border
    (pysfm_amd/csrc/ba_border.h)."""
    from pysfm_amd import synthetic_data as sd
    nc = len(s['R0'])
    rs = np.random.RandomState(seed)
    first = np.sort(rs.choice(np.arange(1, nc // 2 - 1), n, replace=False))
    return sd.add_loop_closure_tracks(s, [(int(i), int(i) + nc // 2) for i in first])


PASS_KERNELS = ('linearize', 'camera_blocks', 'point_invert', 'schur_init', 'schur_pairs')
OTHER_CONFIGS = [   # label, BASELINE config, sensor, outliers, shuffle, drop
    ('config2', 2, 'gaussian', 0., False, 0.),
    ('config4_huber', 4, 'huber', .1, False, 0.),
    ('config4_cauchy', 4, 'cauchy', .1, False, 0.),
    ('config3_shuffled', 3, 'gaussian', 0., True, 0.),
    ('config3_cameras_renumbered', 3, 'gaussian', 0., False, 0., 10, None, 'cameras'),      # the cameras in random order: the library finds the band itself
    ('config3_10_loop_closure_tracks', 3, 'gaussian', 0., False, 0., 10, None, 'loops'),      # camera i and camera i + 500 see the same point, ten times: band + border
    ('config3_cameras_renumbered_10_loop_closure_tracks', 3, 'gaussian', 0., False, 0., 10, None, 'loops+cameras'),      # both at once: the ordering leaves the weak ties out, the border takes them
    ('config3_30pct_dropped', 3, 'gaussian', 0., False, .3),
    ('config3_2pct_tracks_of_80_cameras', 3, 'gaussian', 0., False, 0., 10, (50, 80)),      # a few long tracks: pairs of 32-camera segments on the matrix cores, half-bandwidth 79
    ('config5_one_gpu', 5, 'gaussian', 0., False, 0.),
    ('config5_10_loop_closure_tracks', 5, 'gaussian', 0., False, 0., 10, None, 'loops'),
    ('config3_track_length_13', 3, 'gaussian', 0., False, 0., 13),      # nodes of 12 cameras: since round 5 in the one-launch cyclic reduction (0.44 ms before)
    ('config3_track_length_16', 3, 'gaussian', 0., False, 0., 16),      # windows of six tiles a side (two launches of the reduction), cyclic reduction with four kernels per level
    ('config3_track_length_32', 3, 'gaussian', 0., False, 0., 32),      # long tracks: windows of 32 cameras on the matrix cores (k_schur_wide_mfma), cyclic reduction with 192-unknown nodes in device memory
]


def quick_config(device, label, cfg_id, sensor_name, outliers, shuffle, drop, track_len=10, long_tracks=None, variant=None, steps=20, warmup=8, scene_cache=None):
    """`other_configs`: a short run of one of the other BASELINE configurations / scene shapes on this GPU, the same
    complete LM trial per step, timed the same way (no events in the timed window; the per-kernel numbers come from
    bracketed trials before it)."""
    from pysfm_amd import Bundle, BundleAdjuster, sensor_model
    from pysfm_amd import synthetic_data as sd
    import torch
    cfg = CONFIGS[cfg_id]
    nc, nt = cfg['cams'], cfg['points']
    init_mode = 'pose' if cfg_id == 5 else 'params'
    key = (nc, nt, outliers, init_mode, track_len)
    if scene_cache is not None and key in scene_cache:
        s = scene_cache[key]
    else:
        s = sd.generate_banded_scene(nc, nt, track_len=track_len, outlier_frac=outliers, init_mode=init_mode)
        if scene_cache is not None:
            scene_cache.clear()                     # (one scene at a time: config 5 is 400 MB of host arrays)
            scene_cache[key] = s
    if long_tracks:                                 # (every, length): every `every`-th point seen by `length` consecutive cameras
        s = with_long_tracks(s, nc, nt, track_len, long_tracks[0], long_tracks[1])
    if variant in ('loops', 'loops+cameras'):
        s = with_loop_closures(s)
        nt = len(s['X0'])
    if variant in ('cameras', 'loops+cameras'):
        s = with_cameras_renumbered(s)
    obs_cam, obs_pt, obs_z, X0 = scene_variant(s, track_len, shuffle, drop)
    model = {'gaussian': sensor_model.GaussianModel(1.), 'cauchy': sensor_model.CauchyModel(.05),
             'huber': sensor_model.HuberModel(.06)}[sensor_name]
    bundle = Bundle.FromObservations(s['K'], s['R0'], s['t0'], X0, obs_cam, obs_pt, obs_z, sensor_model=model)
    ba = BundleAdjuster(device=device, verbose=False)
    t0 = time.time()
    ba.set_bundle(bundle)
    t_setup_first = time.time() - t0                # (the handle's first problem: device allocations included)
    t0 = time.time()
    ba.set_bundle(bundle)
    t_setup = time.time() - t0
    be = ba.backend
    be.set_option('reuse_linearization', 0)         # a timed step is a COMPLETE trial: nothing kept from the trial before (the end-to-end run below keeps the default)
    one_trial, state = make_trial_runner(ba, be)
    one_trial()                                     # lazy code-object loading
    be.enable_timing(True)
    be.timings(reset=True)
    for _ in range(warmup):
        one_trial()
    tm = {k: v for k, v in be.timings(reset=True).items() if v['launches'] > 0}
    be.enable_timing(False)
    torch.cuda.synchronize()
    state['paths'] = {}
    t0 = time.time()
    for _ in range(steps):
        one_trial()
    torch.cuda.synchronize()
    dt = time.time() - t0
    nobs = len(obs_cam)
    dom = max(tm, key=lambda k: tm[k]['ms']) if tm else None
    kms = {k: v['ms'] / warmup for k, v in tm.items()}
    pass_ms = sum(kms[k] for k in PASS_KERNELS if k in kms)
    info = be.problem_info()
    out = {'workload': 'BASELINE configs[%d]%s: %d cameras / %d points / %d observations, %s sensor model%s%s%s' % (
               cfg_id - 1, ('' if track_len == 10 else ' with track length %d' % track_len) + ('' if not long_tracks else ' and every %d-th point seen by %d cameras' % tuple(long_tracks)), nc, nt, nobs, sensor_name, ' + %.0f %% gross outliers' % (100 * outliers) if outliers else '',
               (', tracks and observations in random order' if shuffle else '') + (', cameras renumbered at random' if variant in ('cameras', 'loops+cameras') else '') + (', plus 10 loop-closure tracks (camera i and camera i + half the sequence)' if variant in ('loops', 'loops+cameras') else ''),
               ', %.0f %% of the observations dropped at random (ragged tracks)' % (100 * drop) if drop else ''),
           'init_mode': init_mode, 'steps': steps, 'warmup': warmup + 1, 'ms_per_step': 1e3 * dt / steps, 'obs_per_s': nobs * steps / dt,
           'dominant_kernel': None if dom is None else KERNEL_NAMES.get(dom, 'k_' + dom), 'dominant_kernel_ms_per_step': None if dom is None else kms[dom],
           'kernel_ms_per_step': kms, 'linearise_schur_pass_ms': pass_ms,
           'linearise_schur_pass_fraction_of_kernel_time': pass_ms / max(1e-12, sum(kms.values())),
           'obs_jacobians_per_s': nobs / max(1e-9, pass_ms * 1e-3),
           'schur_kernel': info.get('schur_kernel'), 'half_bandwidth': be.half_bandwidth, 'solve_kind': getattr(be, 'last_solve_kind', None),
           'cameras_permuted': info.get('cameras_permuted'), 'caller_half_bandwidth': info.get('caller_half_bandwidth'), 'border_cameras': info.get('border_cameras'),
           'trials_by_solver_and_outcome': dict(state['paths']), 'set_bundle_s': t_setup, 'set_bundle_first_s': t_setup_first}
    # what a caller feels: BundleAdjuster.set_bundle + optimize(25 steps), the adjusted bundle back on the host (library defaults:
    # a trial that follows a rejected one reuses the linearisation of the unchanged current set)
    be.set_option('reuse_linearization', 1)
    torch.cuda.synchronize()
    t0 = time.time()
    ba.set_bundle(bundle)
    ba.optimize(max_steps=25)
    _ = ba.bundle
    torch.cuda.synchronize()
    out['end_to_end_optimize_s'] = time.time() - t0
    out['end_to_end_lm_trials'] = int(ba.lm_trials)
    out['end_to_end_linearizations_reused'] = be.problem_info().get('linearizations_reused')
    be.close()
    return out


def small_problems():
    """The reference's own scale: BASELINE configs[0] (5 cameras x 50 tracks) and the sliding-window caller (window_slam.py:17-48,
    one 10-camera x 100-track problem per frame, 91 frames).  At this size optimize() is ONE resident launch
    (pysfm_amd/csrc/ba_resident.h); `python_loop` is the same call with the loop over ba_lm_trial in Python."""
    from pysfm_amd import Bundle, BundleAdjuster, sensor_model, synthetic_data as sd, window_slam
    out = {}
    s = sd.generate_banded_scene(5, 50, track_len=5, init_perturbation=.03)
    b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], sensor_model=sensor_model.GaussianModel(1.))
    for name, resident in (('resident', True), ('python_loop', False)):
        ba = BundleAdjuster(verbose=False)
        ba.resident = resident
        ba.set_bundle(b)
        ba.optimize()
        t0 = time.time()
        for _ in range(20):
            ba.set_bundle(b)
            ba.optimize()
            _ = ba.bundle
        out['config1_set_bundle_optimize_bundle_s_' + name] = (time.time() - t0) / 20
        out['config1_lm_trials'] = int(ba.lm_trials)
        ba.backend.close()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tests', 'golden', 'scene_oleg_100x1000.npz')
    if os.path.exists(path):
        g = np.load(path)
        wb = Bundle.FromObservations(g['K'], g['R'].reshape(-1, 3, 3), g['t'], g['X'], g['obs_cam'], g['obs_pt'], g['obs_z'],
                                     sensor_model=sensor_model.GaussianModel(1.))
        try:
            for name, resident in (('resident', True), ('python_loop', False)):
                BundleAdjuster.resident = resident
                window_slam.run(wb, 10, num_tracks=100, max_steps=3, verbose=False)
                t0 = time.time()
                _, hist = window_slam.run(wb, 10, num_tracks=100, verbose=False)
                out['window_slam_91_windows_10x100_s_' + name] = time.time() - t0
                out['window_slam_accepted_steps'] = int(sum(len(h) for h in hist) - len(hist))
        finally:
            BundleAdjuster.resident = True
    return out


def unordered_collection(nc=5000, nt=200000, partners=8, track_len=3):
    """A scene with no band under any camera order (synthetic_data.generate_collection_scene: every camera shares tracks with cameras
    drawn at random from all the others): the reduced system is solved by conjugate gradients over the blocks the tracks define
    (csrc/ba_pcg.h).  Round 5: LU down a band 4500 cameras wide, 34.5 s a trial."""
    import torch
    from pysfm_amd import Bundle, BundleAdjuster, synthetic_data as sd
    from pysfm_amd._capi import PARAMS_CUR
    s = sd.generate_collection_scene(nc, nt, partners=partners, track_len=track_len)
    b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
    ba = BundleAdjuster(verbose=False)
    ba.set_bundle(b)
    t0 = time.time()
    ba.set_bundle(b)
    t_setup = time.time() - t0
    be = ba.backend
    cur = ba._cost(PARAMS_CUR)
    ba.trial(10., None, cur)                           # (the first solve of a problem builds the list of blocks)
    out = {'cameras': nc, 'points': nt, 'observations': int(be.nobs), 'half_bandwidth': int(be.half_bandwidth), 'set_bundle_s': t_setup, 'trials': {}}
    for damping in (10., 1., .1, .01):
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5):
            acc, nxt = ba.trial(damping, None, cur)
        torch.cuda.synchronize()
        info = be.pcg_info()
        out['trials']['damping_%g' % damping] = {'ms_per_trial': 1e3 * (time.time() - t0) / 5, 'accepted': bool(acc), 'solver': be.last_solve_kind,
                                                 'iterations': info['iterations'], 'rel_residual': info['rel_residual']}
    out['blocks_upper'] = info['blocks']
    out['band_fill'] = info['band_fill']
    be.enable_timing(True)
    be.timings(reset=True)
    for _ in range(3):
        ba.trial(10., None, cur)
    out['kernel_ms_per_step'] = {k: v['ms'] / 3 for k, v in be.timings(reset=True).items() if v['launches']}
    be.enable_timing(False)
    t0 = time.time()
    ba.set_bundle(b)
    ba.optimize(max_steps=10)
    torch.cuda.synchronize()
    out['optimize_10_steps_s'] = time.time() - t0
    out['optimize_costs'] = [float(c) for c in ba.costs]
    out['optimize_trials'] = [(d, o) for d, o, _ in ba.trial_log]
    ba.backend.close()
    return out


# the reference itself (lib2to3 translation, 1 core Xeon 2.1 GHz) on its own dataset, data/oleg_synthetic
REFERENCE_ON_OLEG = {'prepare_schur_complement_s': 3.54, 'compute_schur_complement_s': 24.65, 'solve_s': 0.0145, 'backsubstitute_s': 0.14,
                     'compute_cost_s': 0.92, 'compute_update_s': 28.3, 'hardware': '1 core Intel Xeon 2.1 GHz (the survey container)',
                     'source': 'BASELINE.md, "Reference path measured during the survey" (bundle_adjuster.py:165-331 on data/oleg_synthetic)'}


def reference_dataset():
    """The ONE input on which a number of the reference exists: its own data/oleg_synthetic (100 cameras x 1000 tracks, every
    track seen by every camera, 100 000 observations; committed as tests/golden/scene_oleg_100x1000.npz together with the
    reference's own compute_update(10.) on it).  compute_update(10.) and optimize(max_steps=10) as the reference's callers
    run them (batch_ba.py:34-35, test_bundle.py:252-253), wall clock and per kernel, beside BASELINE.md's timings of the
    reference, and the update checked against the reference's."""
    from pysfm_amd import Bundle, BundleAdjuster, sensor_model
    path = os.path.join(ROOT, 'tests', 'golden', 'scene_oleg_100x1000.npz')
    if not os.path.exists(path):
        return {'error': 'fixture missing: ' + path}
    g = np.load(path)
    model = sensor_model.GaussianModel(1.) if int(g['sensor_kind']) == 0 else sensor_model.CauchyModel(float(g['sensor_sigma']))
    b = Bundle.FromObservations(g['K'], g['R'].reshape(-1, 3, 3), g['t'], g['X'], g['obs_cam'], g['obs_pt'], g['obs_z'], sensor_model=model)
    import torch
    ba = BundleAdjuster(verbose=False)
    ba.set_bundle(b)
    be = ba.backend
    ba.compute_update(10.)                                # warm-up: code objects, work lists
    torch.cuda.synchronize()
    reps = 10
    t0 = time.time()
    for _ in range(reps):
        mu, su = ba.compute_update(10.)
    torch.cuda.synchronize()
    t_update = (time.time() - t0) / reps
    dC_ref = np.asarray(g['l10_dC'])
    err = float(np.max(np.abs(-mu - dC_ref)) / np.max(np.abs(dC_ref)))
    t0 = time.time()
    for _ in range(reps):
        c = ba.compute_cost(ba.bundle)
    t_cost = (time.time() - t0) / reps
    be.enable_timing(True)
    be.timings(reset=True)
    for _ in range(reps):
        ba.compute_update(10.)
    tm = {k: v['ms'] / reps for k, v in be.timings(reset=True).items() if v['launches'] > 0}
    be.enable_timing(False)
    ba.set_bundle(b)
    ba.optimize(max_steps=10)
    t0 = time.time()
    ba.set_bundle(b)
    ba.optimize(max_steps=10)
    _ = ba.bundle
    torch.cuda.synchronize()
    t_opt = time.time() - t0
    trials = int(ba.lm_trials)
    tb = Bundle.FromObservations(g['K'], g['R'].reshape(-1, 3, 3), g['t'], g['X'], g['obs_cam'], g['obs_pt'], g['obs_z'], sensor_model=model)
    tb.triangulate_all()
    t0 = time.time()
    tb.triangulate_all()
    t_tri = time.time() - t0
    out = {'workload': "the reference's own dataset data/oleg_synthetic: 100 cameras x 1000 tracks, dense visibility, 100 000 observations "
                       '(tests/golden/scene_oleg_100x1000.npz), Gaussian model, camera 0 frozen',
           'compute_update_s': t_update, 'compute_update_obs_per_s': len(g['obs_cam']) / t_update,
           'compute_update_max_rel_diff_to_the_reference': err, 'compute_cost_s': t_cost, 'cost': float(c), 'cost_of_the_reference': float(g['l10_cost']),
           'kernel_ms_per_compute_update': tm, 'schur_kernel': be.problem_info().get('schur_kernel'), 'half_bandwidth': be.half_bandwidth,
           'solve_kind': getattr(be, 'last_solve_kind', None),
           'set_bundle_optimize_10_steps_bundle_s': t_opt, 'optimize_lm_trials': trials, 'optimize_steps': int(ba.num_steps),
           'optimize_costs': [float(ba.costs[0]), float(ba.costs[-1])], 'triangulate_all_s': t_tri,
           'reference': REFERENCE_ON_OLEG, 'speedup_of_compute_update_over_the_reference': REFERENCE_ON_OLEG['compute_update_s'] / t_update}
    be.close()
    return out


def live_pmc_traffic(argv, kernels, timeout_s=150):
    """`roofline.traffic` measured in THIS run: two rocprofv3 passes of this same command (`--pmc FETCH_SIZE`, then
    `--pmc WRITE_SIZE`: the two do not fit one pass; counters only, no trace domains), a few trials each, and per kernel
    HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md, HBM section: KB units, and on gfx950
    FETCH_SIZE reads half of a wide coalesced stream).  Returns ({timer id: bytes per launch}, note); ({}, reason) when
    rocprofv3 is absent or a pass fails - the caller then falls back to the committed summary and says so."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which('rocprofv3')
    if exe is None:
        return {}, 'rocprofv3 not on PATH'
    tmp = tempfile.mkdtemp(prefix='ba_pmc_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp')
    agg = {}
    try:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            d = os.path.join(tmp, counter)
            cmd = [exe, '--pmc', counter, '--output-format', 'csv', '-d', d, '-o', 'pmc', '--', sys.executable,
                   os.path.abspath(__file__)] + argv + ['--pmc-child']
            r = subprocess.run(cmd, cwd='/tmp', env=env, capture_output=True, text=True, timeout=timeout_s)
            files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
            if r.returncode != 0 or not files:
                return {}, 'rocprofv3 --pmc %s failed (rc %d): %s' % (counter, r.returncode, (r.stderr or r.stdout)[-300:])
            for row in csv.DictReader(open(files[0])):
                if row.get('Counter_Name') != counter:
                    continue
                name = row['Kernel_Name'].split('(')[0].replace('void ', '').split('<')[0].split('::')[-1]
                a = agg.setdefault(name, {'FETCH_SIZE': 0., 'WRITE_SIZE': 0., 'n': 0})
                a[counter] += float(row['Counter_Value'])
                if counter == 'FETCH_SIZE':
                    a['n'] += 1
    except Exception as e:          # a time-out, a parse error: the bench line must still come out
        return {}, 'live PMC pass failed: %r' % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {}
    for timer, names in kernels.items():
        for name in names:
            if name in agg and agg[name]['n'] > 0:
                a = agg[name]
                out[timer] = int((2 * a['FETCH_SIZE'] + a['WRITE_SIZE']) / a['n'] * 1024)
                break
    return out, 'live: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of this command inside this run'


# timer id -> kernel names (without template arguments) that run under it, most specific first
PMC_KERNEL_NAMES = {'schur_pairs': ['k_schur_groups_mfma2', 'k_schur_groups_mfma3', 'k_schur_wide_mfma', 'k_schur_rect_mfma', 'k_schur_groups_mfma', 'k_schur_groups', 'k_schur_pairs'],
                    'linearize': ['k_linearize_groups_trial', 'k_linearize_groups', 'k_linearize'], 'backsub': ['k_backsub_groups', 'k_backsub'],
                    'point_invert': ['k_point_invert_schur_init', 'k_point_invert'],
                    'bcr_eliminate': ['k_bcr_eliminate_fused', 'k_bcr_eliminate_split', 'k_bcr_eliminate'],
                    'bcr_backsolve': ['k_bcr_backsolve_fused', 'k_bcr_backsolve'], 'bcr_assemble': ['k_bcr_assemble'],
                    'camera_blocks': ['k_camera_blocks'], 'cost': ['k_cost'], 'schur_init': ['k_schur_init']}


CONFIGS = {2: dict(cams=100, points=10000), 3: dict(cams=1000, points=100000),
           4: dict(cams=1000, points=100000, sensor='huber', outliers=.1), 5: dict(cams=10000, points=1000000, strong=True)}


PHASE = ['start']      # what this rank is doing (the watchdog of multi-GPU runs reports it)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=20,
                    help='untimed trials before the timed ones (the first ~20 run 2-3 %% slower: clocks and caches settling)')
    ap.add_argument('--config', type=int, default=3, choices=sorted(CONFIGS))
    ap.add_argument('--cams', type=int, default=None, help='override the number of cameras of the configuration')
    ap.add_argument('--pts-per-gpu', type=int, default=None, help='override: points per GPU (weak scaling)')
    ap.add_argument('--track-len', type=int, default=10)
    ap.add_argument('--shuffle-points', action='store_true')
    ap.add_argument('--shuffle-cameras', action='store_true', help='renumber the cameras at random: the library orders the optimised cameras itself')
    ap.add_argument('--loop-closures', type=int, default=0, metavar='N', help='add N tracks that tie a camera to the camera half a sequence further on (band + border)')
    ap.add_argument('--drop-observations', type=float, default=0., metavar='FRAC',
                    help='drop this fraction of the observations at random (every track keeps two): camera lists no longer repeat')
    ap.add_argument('--long-tracks', default=None, metavar='EVERY,LENGTH',
                    help='every EVERY-th point is seen by LENGTH consecutive cameras instead of --track-len (features that survive for a long '
                         'stretch of the video): e.g. 50,80')
    ap.add_argument('--sensor', default=None, choices=['gaussian', 'cauchy', 'huber'])
    ap.add_argument('--outliers', type=float, default=None)
    ap.add_argument('--windows', type=int, default=5, help='extra timed windows of --steps trials after the headline one (min / median)')
    ap.add_argument('--force-comm', action='store_true', help='run the sharded path with a one-rank RCCL group on one GPU')
    ap.add_argument('--ranks-on-one-gpu', action='store_true',
                    help='N > 1 without N GPUs: every rank on GPU 0, a gloo process group, the collectives through torch.distributed (staged through the '
                         'host).  NOT a scaling measurement - it exercises the rank code of this file end to end (spawn, rendezvous, sharding, the band width '
                         'agreed over the ranks, the distributed solve, the watchdog, the JSON line); `value` then says what one GPU does with N processes on it')
    ap.add_argument('--collectives', default='library', choices=['library', 'torch'])
    ap.add_argument('--distributed-solve', default='auto', choices=['auto', 'on', 'off'],
                    help='N > 1: spread the reduced camera solve over the ranks (three small sums per trial) instead of summing the whole '
                         'band and solving it on every rank; auto = when the band is larger than 4 MB (config 5)')
    ap.add_argument('--option', action='append', default=[], metavar='NAME=VALUE',
                    help='library option for experiments (HipBackend.set_option), e.g. --option solver=bcr1; recorded in the JSON line')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-table', action='store_true',
                    help='no HIP events at all in the timed region (for external kernel traces); roofline uses the warm-up timings')
    ap.add_argument('--no-lm', action='store_true', help='skip the untimed full optimize() that yields the RMSE')
    ap.add_argument('--init-mode', default=None, choices=['params', 'pose'],
                    help="initial guess of the scene generator: 'params' = Camera.perturb on the raw parameters (SURVEY 8d / the "
                         "reference's test_bundle.py:165-167; default for configs 2-4), 'pose' = the camera turned about its own "
                         "centre (default for config 5, where 'params' throws cameras 20 units off and no LM run recovers)")
    ap.add_argument('--no-other-configs', action='store_true', help='skip the short runs of the other configurations (`other_configs`)')
    ap.add_argument('--no-live-pmc', action='store_true', help='do not run the two rocprofv3 --pmc passes that measure `roofline.traffic` live')
    ap.add_argument('--rank-timeout', type=float, default=900., metavar='SECONDS',
                    help='multi-GPU runs: a rank that has not finished after this long prints what it was doing and exits (rank 0: a JSON line with "error")')
    ap.add_argument('--detail-out', default=None, metavar='PATH',
                    help='where the full record goes (default bench_detail.json beside this file, and a copy under gpurun_out/ when that exists); '
                         'the last stdout line is the short headline the driver parses')
    ap.add_argument('--full-line', action='store_true',
                    help='experiments (scripts/ab_*.sh): print the WHOLE record as the last line instead of the short headline')
    ap.add_argument('--pmc-child', action='store_true', help=argparse.SUPPRESS)       # the run rocprofv3 wraps: a few trials, no JSON line
    args = ap.parse_args()
    global PMC_WORKLOAD
    plain = args.cams is None and args.pts_per_gpu is None and args.track_len == 10 and not args.option and not args.drop_observations and not args.long_tracks and not args.shuffle_cameras and not args.loop_closures
    PMC_WORKLOAD = ('config3' if args.config == 3 else 'config%d' % args.config) if plain and args.gpus == 1 and args.sensor is None and args.outliers is None else 'none'

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        spawn_ranks(args.gpus, args.ranks_on_one_gpu)      # does not return
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.ranks_on_one_gpu:
        local_rank = 0
    if world != args.gpus:
        sys.stderr.write('bench.py: --gpus %d but WORLD_SIZE=%d\n' % (args.gpus, world))
        sys.exit(2)

    import torch
    from pysfm_amd import Bundle, BundleAdjuster, sensor_model
    from pysfm_amd import synthetic_data as sd
    from pysfm_amd._capi import PARAMS_CUR

    if world > 1:
        # a rank stuck in a collective (a communicator that never forms, a peer that died) must not become a silent driver
        # time-out: after --rank-timeout seconds every rank says where it was and leaves; rank 0 leaves a JSON line with "error"
        import threading

        def give_up():
            msg = 'bench.py rank %d/%d: no result after %.0f s, last phase: %s' % (rank, world, args.rank_timeout, PHASE[0])
            sys.stderr.write(msg + '\n')
            sys.stderr.flush()
            if rank == 0:
                print(json.dumps({'metric': 'LM-iter throughput (obs/sec) + final reproj RMSE, 1k-cam/100k-pt/1M-obs scene', 'value': None, 'unit': 'obs/s',
                                  'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'error': msg}), flush=True)
            os._exit(3)
        watchdog = threading.Timer(args.rank_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()

    assert torch.cuda.is_available(), 'bench.py needs an MI355X (no CPU path)'
    comm = None
    if world > 1 or args.force_comm:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', str(free_port()))
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        torch.cuda.set_device(local_rank)
        PHASE[0] = 'torch.distributed.init_process_group(nccl)'
        if args.ranks_on_one_gpu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        from pysfm_amd.distributed import ShardComm, shard_tracks
        PHASE[0] = 'ShardComm: agreeing on the collectives (%s)' % args.collectives
        comm = ShardComm(collectives='torch' if args.ranks_on_one_gpu else args.collectives)
        comm.local_rank = local_rank
    ngpus = world

    # ---- scene (identical on every rank), then this rank's shard of the tracks
    cfg = CONFIGS[args.config]
    nc = args.cams or cfg['cams']
    strong = bool(cfg.get('strong')) and args.pts_per_gpu is None
    nt = cfg['points'] if strong else (args.pts_per_gpu or cfg['points']) * ngpus
    sensor_name = args.sensor or cfg.get('sensor', 'gaussian')
    outliers = cfg.get('outliers', 0.) if args.outliers is None else args.outliers
    init_mode = args.init_mode or ('pose' if args.config == 5 else 'params')
    s = sd.generate_banded_scene(nc, nt, track_len=args.track_len, outlier_frac=outliers, init_mode=init_mode)
    if args.long_tracks:
        every, llen = [int(v) for v in args.long_tracks.split(',')]
        s = with_long_tracks(s, nc, nt, args.track_len, every, llen)
    if args.shuffle_cameras:
        s = with_cameras_renumbered(s)
    if args.loop_closures:
        s = with_loop_closures(s, args.loop_closures)
        nt = len(s['X0'])
    obs_cam, obs_pt, obs_z, X0 = scene_variant(s, args.track_len, args.shuffle_points, args.drop_observations) if not args.long_tracks \
        else (s['obs_cam'], s['obs_pt'], s['obs_z'], s['X0'])
    model = {'gaussian': sensor_model.GaussianModel(1.), 'cauchy': sensor_model.CauchyModel(.05),
             'huber': sensor_model.HuberModel(.06)}[sensor_name]
    bundle = Bundle.FromObservations(s['K'], s['R0'], s['t0'], X0, obs_cam, obs_pt, obs_z, sensor_model=model)
    ba = BundleAdjuster(device=local_rank, comm=comm, verbose=False)
    for kv in args.option:
        name, _, val = kv.partition('=')
        ba.backend.set_option(name, val)
    track_ids = None
    if comm is not None:
        if args.distributed_solve != 'auto':
            ba.distributed_solve = args.distributed_solve == 'on'
        # (cut where the distributed reduced solve wants the tracks cut, when it is going to be used; balanced by observations otherwise)
        use_plan = args.distributed_solve == 'on' or (args.distributed_solve == 'auto' and args.config == 5 and world > 1)
        track_ids = shard_tracks(bundle, rank, world, plan=ba.backend.dist_plan if use_plan else None)
    PHASE[0] = 'set_bundle (ba_comm_init with the library collectives, the band width agreed over the ranks)'
    t_setup_first = time.time()
    ba.set_bundle(bundle, track_ids=track_ids)
    t_setup_first = time.time() - t_setup_first     # (the handle's first problem: library start-up and device allocations included)
    t_setup = time.time()
    ba.set_bundle(bundle, track_ids=track_ids)
    t_setup = time.time() - t_setup
    be = ba.backend
    nobs_local = be.nobs
    nobs_total = len(obs_cam)

    def sync():
        torch.cuda.synchronize()
        if comm is not None:
            comm.barrier()
            torch.cuda.synchronize()

    # ---- (untimed) the full LM run: final cost + reprojection RMSE
    lm = {}
    if not args.no_lm:
        PHASE[0] = 'the untimed optimize(25 steps)'
        sync()
        t0 = time.time()
        ba.optimize(max_steps=25)
        sync()
        lm_wall = time.time() - t0
        e = be.eval_observations(PARAMS_CUR, e=True, r=False, Jc=False, Jp=False)['e']
        sq, cnt = float(np.sum(e * e)), float(len(e))
        if comm is not None:
            sq, cnt = comm.allreduce_scalar(sq), comm.allreduce_scalar(cnt)
        # "next to the CPU restatement's value on the same scene and seed" - the oracle's own 25-step walk of this
        # scene (oracle/gen_golden_lm25.py ran it in the build container, ~25 minutes; tests/golden/<name>_lm25.npz)
        oracle_run = None
        golden = {(3, 'gaussian', 'params'): 'config3', (3, 'gaussian', 'pose'): 'config3_pose', (4, 'huber', 'params'): 'config4_huber'}.get((args.config, sensor_name, init_mode))
        if golden and plain and not args.shuffle_points:
            path = os.path.join(ROOT, 'tests', 'golden', golden + '_lm25.npz')
            if os.path.exists(path):
                gd = np.load(path)
                oracle_run = dict(final_reproj_rmse=float(gd['rmse_final']), initial_reproj_rmse=float(gd['rmse_initial']), lm_steps=int(gd['num_steps']),
                                  lm_trials=int(len(gd['trial_damping'])), lm_converged=bool(gd['converged']), lm_cost_initial=float(gd['costs'][0]),
                                  lm_cost_final=float(gd['costs'][-1]), oracle_wall_s=float(gd['oracle_wall_s']), source='tests/golden/%s_lm25.npz' % golden,
                                  note='oracle/ba_oracle.py lm_optimize (NumPy restatement of bundle_adjuster.py:117-162) on the same scene and seed, run in the build '
                                       'container; the two walks take the same decisions up to the first step accepted at a damping below 1e-2 and are two samples '
                                       'of an ill-conditioned sequence after it (tests/test_gpu_configs.py::test_full_lm25_run_against_the_oracle_golden_walk)')
        lm = dict(final_reproj_rmse=float(np.sqrt(sq / cnt)), final_reproj_rmse_oracle=None if oracle_run is None else oracle_run['final_reproj_rmse'],
                  oracle_lm_run=oracle_run, lm_steps=ba.num_steps,
                  lm_trials=int(ba.lm_trials), lm_converged=bool(ba.converged),
                  lm_cost_initial=ba.costs[0], lm_cost_final=ba.costs[-1], lm_wall_s=lm_wall,
                  lm_cholesky_rejections=int(getattr(ba, 'cholesky_rejections', 0)))
        if comm is None and nt <= 150000 and args.init_mode is None and not args.pmc_child:
            # the same LM run from the generator's other initial guess (round 1 / SURVEY 8d used 'params', round 2 'pose')
            other = 'pose' if init_mode == 'params' else 'params'
            s2 = sd.generate_banded_scene(nc, nt, track_len=args.track_len, outlier_frac=outliers, init_mode=other)
            if args.shuffle_cameras:
                s2 = with_cameras_renumbered(s2)
            if args.loop_closures:
                s2 = with_loop_closures(s2, args.loop_closures)
            oc2, op2, oz2, X02 = scene_variant(s2, args.track_len, args.shuffle_points, args.drop_observations)
            ba.set_bundle(Bundle.FromObservations(s2['K'], s2['R0'], s2['t0'], X02, oc2, op2, oz2, sensor_model=model))
            ba.optimize(max_steps=25)
            e2 = ba.backend.eval_observations(PARAMS_CUR, e=True, r=False, Jc=False, Jp=False)['e']
            lm['lm_other_start'] = dict(init_mode=other, final_reproj_rmse=float(np.sqrt(np.sum(e2 * e2) / len(e2))), lm_steps=ba.num_steps,
                                        lm_trials=int(ba.lm_trials), lm_converged=bool(ba.converged), lm_cost_initial=ba.costs[0],
                                        lm_cost_final=ba.costs[-1])
            del s2
        # restart from the initial guess for the timed trials
        ba.set_bundle(bundle, track_ids=track_ids)
        be = ba.backend

    # ---- timed region: K complete LM trials, continuing the LM schedule
    PHASE[0] = 'warm-up and timed trials'
    # a timed step is a COMPLETE trial, nothing kept from the trial before: the library's reuse of the linearisation after a
    # rejected trial (default on; it is what end_to_end_optimize_s and the untimed LM run above have) is switched off here
    be.set_option('reuse_linearization', 0)
    one_trial, state = make_trial_runner(ba, be)
    if args.pmc_child:
        # the run rocprofv3 --pmc wraps (live_pmc_traffic): a few complete trials, nothing printed
        for _ in range(max(1, args.warmup) + max(1, args.steps)):
            one_trial()
        torch.cuda.synchronize()
        be.close()
        return

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
    # microseconds of stream time - bracketing all ~25 launches of a 0.3 ms step would slow it ~15 %)
    ev_stride = max(4, args.steps // 2)                 # (20 steps: steps 0 and 10) - a sampled step costs ~45 us of stream and host time: two samples of a kernel whose duration varies by 0.5 % are enough
    be.enable_timing(not args.no_kernel_table, only=[dom], stride=ev_stride)
    sync()
    state['paths'] = {}
    refined0 = be.problem_info().get('solves_refined', 0)
    t0 = time.time()
    for _ in range(args.steps):
        one_trial()
    sync()
    dt = time.time() - t0
    timed_refined = be.problem_info().get('solves_refined', 0) - refined0      # (trials damped below 1e-2 take the solve's refinement step: +39 us each)
    timed_paths = dict(state['paths'])
    tm_dom = be.timings(reset=True)[dom]
    # further windows of the same length, no events at all: spread of the headline number
    be.enable_timing(False)
    window_ms = []
    for _ in range(max(0, args.windows)):
        sync()
        tw = time.time()
        for _ in range(args.steps):
            one_trial()
        sync()
        wdt = time.time() - tw
        if comm is not None:
            wdt = comm_max(comm, wdt)
        window_ms.append(1e3 * wdt / args.steps)
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
        nco, hb = be.nco, be.half_bandwidth
        BORDER_CAMS[0] = int(be.problem_info().get('border_cameras', 0))
        def ab(k):
            n = (ours[k]['launches'] / nprof) if k in ours else 1.
            v = algorithmic_bytes(k, be.nc, nco, be.nt, nobs_local, be.nt, hb, launches=n)
            if k == 'bcr_eliminate' and 'bcr_backsolve' not in ours:      # the back-substitution items ride in the elimination's launch
                v += algorithmic_bytes('bcr_backsolve', be.nc, nco, be.nt, nobs_local, be.nt, hb, launches=n)
            return v
        avg_ms = tm_dom['ms'] / max(1, tm_dom['launches'])
        B = ab(dom)
        achieved = B / (avg_ms * 1e-3) / 1e9
        # HBM traffic of the kernels from the PMC counters: measured live (two rocprofv3 --pmc passes of this same
        # workload, a few trials each) when rocprofv3 is there; else from the newest committed summary of this workload,
        # flagged as possibly stale
        live, live_note = {}, 'not attempted (--no-live-pmc, N > 1, or a communicator attached)'
        if ngpus == 1 and comm is None and not args.no_live_pmc:
            child = ['--config', str(args.config), '--track-len', str(args.track_len), '--steps', '4', '--warmup', '2', '--windows', '0',
                     '--no-lm', '--no-cpu-baseline', '--no-other-configs', '--no-live-pmc', '--init-mode', init_mode]
            child += ['--cams', str(args.cams)] if args.cams else []
            child += ['--pts-per-gpu', str(args.pts_per_gpu)] if args.pts_per_gpu else []
            child += ['--shuffle-points'] if args.shuffle_points else []
            child += ['--shuffle-cameras'] if args.shuffle_cameras else []
            child += ['--loop-closures', str(args.loop_closures)] if args.loop_closures else []
            child += ['--drop-observations', str(args.drop_observations)] if args.drop_observations else []
            child += ['--sensor', args.sensor] if args.sensor else []
            child += ['--outliers', str(args.outliers)] if args.outliers is not None else []
            for kv in args.option:
                child += ['--option', kv]
            t_pmc = time.time()
            live, live_note = live_pmc_traffic(child, PMC_KERNEL_NAMES)
            live_note += ' (%.0f s)' % (time.time() - t_pmc)

        def traffic_of(k):
            if k in live:
                return live[k], 'live'
            return pmc_traffic(k)
        traffic, traffic_src = traffic_of(dom)
        copy_gbs = be.measure_copy_bandwidth()
        sflops = schur_flops(nobs_local, be.nt)
        # the whole reduction of one trial (one launch up to track length 13, two to four beyond: DESIGN.md)
        schur_ms = ours['schur_pairs']['ms'] / nprof if 'schur_pairs' in ours else None
        limited = KERNEL_BOUND.get(dom, 'hbm')
        # Which roof binds: the kernel's arithmetic intensity (useful flops of its launches in one trial / the bytes they must
        # move) against the fp64 ridge, 78.6 TF / 8 TB/s = 9.8 flop/B.  Above it `frac` is priced against the fp64 matrix /
        # vector peak ("mfma"), below against HBM; the other roof's numbers ride along (hbm_* / fp64_*), and `limited_by`
        # says what the counters and time lines show actually limits the kernel (profiles/r05_sq_counters.csv).
        n_launch = max(1., (ours[dom]['launches'] / nprof) if dom in ours else 1.)
        step_ms = avg_ms * n_launch                                     # this kernel's launches of ONE trial
        flops = useful_flops(dom, be.nc, nco, be.nt, nobs_local, hb)
        if dom == 'bcr_eliminate' and 'bcr_backsolve' not in ours:
            flops += useful_flops('bcr_backsolve', be.nc, nco, be.nt, nobs_local, hb)
        min_bytes = minimal_bytes(dom, be.nc, nco, be.nt, nobs_local, be.nt, hb)
        intensity = flops / max(1., min_bytes)
        tflops = flops / (step_ms * 1e-3) / 1e12
        priced = 'mfma' if intensity > FP64_RIDGE_FLOP_PER_BYTE else 'hbm'
        roof = {'bound': priced, 'limited_by': limited, 'kernel': KERNEL_NAMES.get(dom, 'k_' + dom), 'timer': dom,
                'achieved': tflops if priced == 'mfma' else achieved, 'peak': FP64_MATRIX_PEAK_TFLOPS if priced == 'mfma' else HBM_PEAK_GBS,
                'unit': 'TFLOP/s' if priced == 'mfma' else 'GB/s',
                'frac': tflops / FP64_MATRIX_PEAK_TFLOPS if priced == 'mfma' else achieved / HBM_PEAK_GBS,
                'traffic': traffic, 'traffic_source': traffic_src,
                'traffic_stale_possible': bool(traffic is not None and traffic_src != 'live'), 'traffic_note': live_note,
                'flops': flops, 'frac_fp64': tflops / FP64_MATRIX_PEAK_TFLOPS, 'fp64_achieved_tflops': tflops, 'fp64_peak_tflops': FP64_MATRIX_PEAK_TFLOPS,
                'arithmetic_intensity_flop_per_byte': intensity, 'ridge_flop_per_byte': FP64_RIDGE_FLOP_PER_BYTE,
                'bytes_in_plus_out_per_trial': min_bytes, 'hbm_achieved_GBps': achieved, 'hbm_frac': achieved / HBM_PEAK_GBS,
                'hbm_frac_in_plus_out': min_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                'measured_copy_GBps': copy_gbs, 'frac_of_measured_copy': achieved / copy_gbs,
                'algorithmic_bytes_per_launch': B, 'avg_launch_ms': avg_ms, 'launches': tm_dom['launches'], 'launches_per_trial': n_launch,
                'note': 'HIP events on the launch stream during the timed steps, every %d-th step; back-to-back launches of one kernel ' % ev_stride +
                        'share one event pair, avg = elapsed / launches.  flops = useful fp64 flops of this kernel per trial (bench.py useful_flops); '
                        'algorithmic_bytes_per_launch counts the workspace the algorithm itself writes and re-reads (hbm_achieved_GBps), '
                        'bytes_in_plus_out_per_trial only what must cross HBM.  limited_by: latency = a chain of dependent pivots and hand-overs '
                        'between workgroups, neither roof'}
        if dom == 'schur_pairs' and schur_ms:
            roof.update({'useful_flops_per_reduction': sflops, 'reduction_ms': schur_ms})
        # every kernel of the trial against both roofs
        per_kernel = {}
        for k, v in ours.items():
            kms = v['ms'] / nprof
            if not kms > 0.:
                continue
            kb = ab(k) * (v['launches'] / nprof)
            kf = useful_flops(k, be.nc, nco, be.nt, nobs_local, hb)
            mb = minimal_bytes(k, be.nc, nco, be.nt, nobs_local, be.nt, hb)
            per_kernel[k] = {'ms_per_trial': kms, 'flops': kf, 'bytes': kb, 'bytes_in_plus_out': mb, 'intensity': kf / max(1., mb),
                             'bound': 'mfma' if kf / max(1., mb) > FP64_RIDGE_FLOP_PER_BYTE else 'hbm', 'limited_by': KERNEL_BOUND.get(k, 'hbm'),
                             'GBps': kb / (kms * 1e-3) / 1e9, 'frac_hbm': kb / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             'tflops': kf / (kms * 1e-3) / 1e12, 'frac_fp64': kf / (kms * 1e-3) / 1e12 / FP64_MATRIX_PEAK_TFLOPS}
        roof['per_kernel'] = per_kernel
        # the pass BASELINE's metric counts: linearise + point inversion + Schur reduction
        pass_kernels = [k for k in ('linearize', 'camera_blocks', 'point_invert', 'schur_init', 'schur_pairs') if k in ours]
        pass_ms = sum(ours[k]['ms'] for k in pass_kernels) / nprof
        pass_bytes = 20 * nobs_local + 96 * be.nc + 24 * be.nt + 96 * be.nt + 72 * be.nt + 288 * nco * (hb + 1) + 48 * nco
        # (per launch x launches per trial: the inversion kernel only runs in the trials after a rejected one - a trial that linearises
        #  inverts inside k_linearize_groups_trial - and the reductions of long tracks are several launches)
        pass_traffic = [None if traffic_of(k)[0] is None else traffic_of(k)[0] * (ours[k]['launches'] / nprof) for k in pass_kernels]
        pass_traffic_live = all(k in live for k in pass_kernels)
        pass_traffic = sum(pass_traffic) if pass_traffic and all(t is not None for t in pass_traffic) else None
        out = {
            'metric': 'LM-iter throughput (obs/sec) + final reproj RMSE, 1k-cam/100k-pt/1M-obs scene',
            'value': value, 'unit': 'obs/s', 'n_gpus': ngpus, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'strong' if strong else 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'ms_per_step_windows': {'n': len(window_ms), 'steps_each': args.steps, 'min': min(window_ms) if window_ms else None,
                                    'median': float(np.median(window_ms)) if window_ms else None, 'max': max(window_ms) if window_ms else None,
                                    'note': 'further windows after the headline one, no HIP events in flight'},
            'config': {'workload': 'BASELINE configs[%d]%s: %d cameras / %d points / %d observations (track length %d)%s, full LM trial per step '
                                   '(linearise+damp+pinv+Schur%s+reduced solve+backsub+update+cost), %s sensor model%s, camera 0 frozen%s'
                                   % (args.config - 1, '' if (args.cams is None and args.pts_per_gpu is None and args.track_len == 10) else ' (modified by flags)',
                                      nc, nt, nobs_total, args.track_len,
                                      ' (%s scaling: %d points / %d obs on this GPU)' % ('strong' if strong else 'weak', be.nt, nobs_local) if ngpus > 1 else '',
                                      ('+gloo all-reduce' if args.ranks_on_one_gpu else '+RCCL all-reduce') if comm is not None else '', sensor_name,
                                      ' + %.0f %% gross outliers' % (100 * outliers) if outliers else '',
                                      (', tracks and observations handed over in random order' if args.shuffle_points else '') +
                                      (', cameras renumbered at random' if args.shuffle_cameras else '') +
                                      (', %.0f %% of the observations dropped at random (ragged tracks)' % (100 * args.drop_observations) if args.drop_observations else '')),
                       'cameras': nc, 'points': nt, 'observations': nobs_total, 'track_len': args.track_len,
                       'init_mode': init_mode, 'shuffled': bool(args.shuffle_points),
                       'parallelism': 'points sharded x%d' % ngpus + (' (ALL RANKS ON ONE GPU, gloo: a functional run of the rank code, not a scaling measurement)' if args.ranks_on_one_gpu else ''),
                       'ranks_on_one_gpu': bool(args.ranks_on_one_gpu),
                       'library_options': args.option or None,
                       'collectives': None if comm is None else ('RCCL inside the library (ba_comm_*)' if getattr(be, 'direct_comm', False)
                                                                 else 'torch.distributed (%s): ' % ('gloo, staged through the host' if args.ranks_on_one_gpu else 'RCCL') + str(getattr(comm, 'direct_fallback_reason', None))),
                       'reduced_solve': None if comm is None else (
                           'spread over the ranks: %(cams_per_node)d cameras per node, %(nodes)d nodes, %(nodes_per_rank)d per rank, %(separators)d separators '
                           'eliminated by every rank; three sums per trial (shared band rows, separators + subtree roots, solution)' % be._dist_info
                           if getattr(ba, '_dist', False) else 'whole band summed over the ranks, solved by every rank'),
                       'allreduce_payload_bytes_per_rank_per_trial': None if comm is None else (
                           8 * sum(be._dist_info['exchange%d_doubles' % k] for k in (1, 2, 3)) + 8 * 2050 if getattr(ba, '_dist', False)
                           else 8 * (be.S_doubles + 6 * nco) + 8 * 2050),
                       'band_bytes': 8 * (be.S_doubles + 6 * nco)},
            'roofline': roof,
            'roofline_linearise_schur_pass': {
                'kernels': [KERNEL_NAMES.get(k, 'k_' + k) for k in pass_kernels], 'ms': pass_ms, 'algorithmic_bytes': pass_bytes,
                'achieved_GBps': pass_bytes / max(1e-9, pass_ms * 1e-3) / 1e9, 'frac_of_hbm_peak': pass_bytes / max(1e-9, pass_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                'traffic': pass_traffic, 'traffic_over_algorithmic': None if pass_traffic is None else pass_traffic / pass_bytes,
                'traffic_source': 'live' if pass_traffic_live else 'committed summary (possibly stale)',
                'traffic_per_kernel': {k: traffic_of(k)[0] for k in ours},
                'obs_jacobians_per_s': nobs_local / max(1e-9, pass_ms * 1e-3),
                'flops': LINEARISE_FLOPS_PER_OBS * nobs_local + sflops + 400. * be.nt,
                'arithmetic_intensity_flop_per_byte': (LINEARISE_FLOPS_PER_OBS * nobs_local + sflops + 400. * be.nt) / pass_bytes,
                'bound': 'mfma' if (LINEARISE_FLOPS_PER_OBS * nobs_local + sflops + 400. * be.nt) / pass_bytes > FP64_RIDGE_FLOP_PER_BYTE else 'hbm',
                'achieved_tflops': (LINEARISE_FLOPS_PER_OBS * nobs_local + sflops + 400. * be.nt) / max(1e-9, pass_ms * 1e-3) / 1e12,
                'frac_fp64': (LINEARISE_FLOPS_PER_OBS * nobs_local + sflops + 400. * be.nt) / max(1e-9, pass_ms * 1e-3) / 1e12 / FP64_MATRIX_PEAK_TFLOPS,
                'note': 'bytes: observations 20/obs + cameras + points + point blocks and inverses (96 + 72 per point) + band S + b, each once; '
                        'W is never materialised'},
            'matrix_cores': {'kernel': 'k_schur_groups_mfma2 / k_schur_groups_mfma3 (timer schur_pairs)', 'useful_flops_per_reduction': sflops,
                             'achieved_tflops': sflops / (schur_ms * 1e-3) / 1e12 if schur_ms else None,
                             'peak_tflops': FP64_MATRIX_PEAK_TFLOPS,
                             'note': 'fp64 MFMA is used where the path is GEMM-shaped (Schur reduction, cyclic-reduction nodes)'},
            'set_bundle_s': t_setup,          # id bookkeeping (host) + internal order, work lists (device + O(points) on the host) + upload (outside the timed region)
            'set_bundle_first_s': t_setup_first,
            'problem_info': be.problem_info(),
            'kernel_ms_per_step': {k: v['ms'] / nprof for k, v in ours.items()},
            'kernel_launches_per_step': {k: v['launches'] / nprof for k, v in ours.items()},
            'all_kernels': {'algorithmic_bytes_per_step': int(sum(ab(k) * v['launches'] for k, v in ours.items()) / nprof),
                            'kernel_ms_per_step': sum(v['ms'] for v in ours.values()) / nprof,
                            'note': 'measured on %d extra trials outside the timed region, every kernel bracketed' % nprof},
            'reduced_system': {'cameras_optimised': nco, 'block_half_bandwidth': hb,
                               'bytes': 8 * be.S_doubles, 'solve_path': getattr(be, 'last_solve_path', None),
                               'solve_kind': getattr(be, 'last_solve_kind', None),
                               'timed_trials_by_solver_and_outcome': timed_paths, 'timed_trials_refined': int(timed_refined)},
        }
        out.update(lm)
        if comm is None and not args.no_lm:
            # what a caller feels: BundleAdjuster.set_bundle + optimize(25 steps), the adjusted bundle back on the host
            be.set_option('reuse_linearization', 1)
            torch.cuda.synchronize()
            t0 = time.time()
            ba.set_bundle(bundle, track_ids=track_ids)
            t1 = time.time()
            ba.optimize(max_steps=25)
            t2 = time.time()
            _ = ba.bundle
            torch.cuda.synchronize()
            out['end_to_end_optimize_s'] = time.time() - t0
            out['end_to_end_parts_s'] = {'set_bundle': t1 - t0, 'optimize': t2 - t1, 'bundle_to_host': time.time() - t2, 'lm_trials': int(ba.lm_trials),
                                         'linearizations_reused': be.problem_info().get('linearizations_reused'),
                                         'note': 'library defaults: a trial that follows a rejected one reuses the linearisation of the unchanged current set '
                                                 '(the timed steps above do not: option reuse_linearization = 0 there)'}
        plain3 = (args.config == 3 and args.cams is None and args.pts_per_gpu is None and args.track_len == 10 and not args.option
                  and not args.drop_observations and not args.shuffle_points and not args.shuffle_cameras and not args.loop_closures and args.sensor is None and args.outliers is None and not args.long_tracks)
        if ngpus == 1 and comm is None and plain3 and not args.no_other_configs:
            # the other BASELINE configurations and scene shapes, a short run each on this same GPU (the headline handle is idle)
            t_oc = time.time()
            cache, oc = {}, {}
            for spec in OTHER_CONFIGS:
                try:
                    oc[spec[0]] = quick_config(local_rank, *spec, scene_cache=cache)
                except Exception as e:                     # one failing variant must not take the headline line with it
                    oc[spec[0]] = {'error': repr(e)}
            try:
                oc['reference_dataset_oleg_100x1000'] = reference_dataset()
            except Exception as e:
                oc['reference_dataset_oleg_100x1000'] = {'error': repr(e)}
            try:
                oc['unordered_collection_5000_cameras'] = unordered_collection()
            except Exception as e:
                oc['unordered_collection_5000_cameras'] = {'error': repr(e)}
            oc['wall_s'] = time.time() - t_oc
            out['other_configs'] = oc
            try:
                out['small_problems'] = small_problems()
            except Exception as e:
                out['small_problems'] = {'error': repr(e)}
        if ngpus == 1 and not args.no_cpu_baseline:
            if nc <= 1500 and nt <= 150000:
                out['cpu_baseline'] = cpu_baseline(s)
            else:       # the oracle's dense S does not fit at this size: bounded sample = the config-3 scene from the same generator
                out['cpu_baseline'] = cpu_baseline(sd.generate_banded_scene(1000, 100000, track_len=args.track_len, outlier_frac=outliers))
                out['cpu_baseline']['sample'] += ' (bounded sample: the 1000-camera / 100 000-point scene of the same generator, not the benched one)'
    if comm is not None:
        import ctypes
        import torch.distributed as dist
        ctypes.CDLL(None).fflush(None)          # (every rank's buffered C output is out before rank 0 prints)
        dist.barrier()
        if getattr(be, 'direct_comm', False):   # the library's own communicator goes before the process group does
            be.synchronize()
            be.comm_detach()
        dist.barrier()
        dist.destroy_process_group()
    # RCCL writes a version banner to the C stdout buffer, which is flushed at exit, i.e. AFTER anything Python
    # prints: flush it now so that the JSON line is the last line of stdout
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if rank == 0:
        detail_path = write_detail(out, args.detail_out)
        print(json.dumps(out if args.full_line else headline(out, detail_path)), flush=True)


HEADLINE_MAX_BYTES = 4096      # the driver reads the LAST stdout line and keeps only a few KB of tail: the line it parses stays short


def write_detail(out, path):
    """Everything the run measured (other configurations, per-kernel rooflines, small problems, problem_info ...) goes to a side
    file; the headline line names it.  A second copy under gpurun_out/ (when that directory exists) travels back from a GPU box."""
    path = path or os.path.join(ROOT, 'bench_detail.json')
    written = None
    for p in [path] + ([os.path.join(ROOT, 'gpurun_out', os.path.basename(path))] if os.path.isdir(os.path.join(ROOT, 'gpurun_out')) else []):
        try:
            with open(p, 'w') as f:
                json.dump(out, f)
                f.write('\n')
            written = written or p
        except OSError as e:
            sys.stderr.write('bench.py: could not write %s: %s\n' % (p, e))
    return written


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def headline(out, detail_path=None):
    """The ONE line the driver parses: the contract's keys, `roofline` and `cpu_baseline` reduced to their numbers, at most
    HEADLINE_MAX_BYTES long.  Everything else is in the detail file (write_detail)."""
    if out is None or 'error' in out:
        return out
    h = _pick(out, ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data'))
    c = out.get('config', {})
    h['config'] = _pick(c, ('workload', 'cameras', 'points', 'observations', 'track_len', 'init_mode', 'parallelism', 'collectives', 'library_options'))
    h['config'] = {k: v for k, v in h['config'].items() if v is not None}
    r = out.get('roofline', {})
    h['roofline'] = _pick(r, ('bound', 'limited_by', 'achieved', 'peak', 'unit', 'frac', 'flops', 'traffic', 'traffic_source', 'traffic_stale_possible',
                              'algorithmic_bytes_per_launch', 'avg_launch_ms', 'launches', 'launches_per_trial', 'hbm_achieved_GBps', 'hbm_frac',
                              'bytes_in_plus_out_per_trial', 'measured_copy_GBps'))
    h['roofline']['kernel'] = str(r.get('kernel', '')).split(':')[0].split(' (')[0][:60]
    p = out.get('roofline_linearise_schur_pass', {})
    h['linearise_schur_pass'] = _pick(p, ('ms', 'obs_jacobians_per_s', 'algorithmic_bytes', 'achieved_GBps', 'frac_of_hbm_peak', 'traffic', 'traffic_over_algorithmic',
                                          'bound', 'achieved_tflops', 'frac_fp64'))
    if out.get('kernel_ms_per_step'):
        h['kernel_us_per_step'] = {k: round(1e3 * v, 2) for k, v in out['kernel_ms_per_step'].items() if v > 0}
    w = out.get('ms_per_step_windows') or {}
    h['ms_per_step_windows'] = _pick(w, ('n', 'min', 'median', 'max'))
    for k in ('final_reproj_rmse', 'final_reproj_rmse_oracle', 'lm_steps', 'lm_trials', 'lm_converged', 'end_to_end_optimize_s', 'set_bundle_s'):
        if k in out:
            h[k] = out[k]
    if 'timed_trials_refined' in out.get('reduced_system', {}):
        h['timed_trials_refined'] = out['reduced_system']['timed_trials_refined']      # of `steps`: those damped below 1e-2, whose solve takes the refinement step
    cb = out.get('cpu_baseline')
    if cb:
        h['cpu_baseline'] = _pick(cb, ('value', 'unit', 'cores', 'kind'))
        h['cpu_baseline']['sample'] = str(cb.get('sample', ''))[:320]
        if 'per_observation_loop' in cb:
            h['cpu_baseline']['per_observation_loop_obs_per_s_1_core'] = cb['per_observation_loop'].get('value')
    if detail_path:
        h['detail'] = os.path.relpath(detail_path, ROOT)
    line = json.dumps(h)
    for drop in ('kernel_us_per_step', 'linearise_schur_pass', 'ms_per_step_windows'):      # never over the limit, whatever a flag adds
        if len(line) <= HEADLINE_MAX_BYTES:
            break
        h.pop(drop, None)
        line = json.dumps(h)
    if len(line) > HEADLINE_MAX_BYTES:
        h['config']['workload'] = h['config']['workload'][:200]
        h.get('cpu_baseline', {}).pop('sample', None)
    return h


def comm_max(comm, x):
    import torch
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device=comm.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


if __name__ == '__main__':
    main()
