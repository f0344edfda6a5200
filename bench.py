#!/usr/bin/env python
"""Benchmark of the RAFT forward-prediction hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: re-runs itself as N ranks under
                                                            torch.distributed.run on 127.0.0.1, see self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): image-pairs/sec at 448x512, iters_pred=24.  A "step" is one forward pass of
RAFT over one batch of synthetic image pairs per GPU (BASELINE configs[1]: batch 4 per GPU,
random Keras-default weights); all 24 upsampled predictions are produced, as the reference does.
Inputs are generated on the device before the timed region.  With N > 1 every rank runs its own
batch (weak scaling) and the final predictions are all-gathered over RCCL inside the timed region.

N = 1 runs BASELINE configs[1] (batch 4 on the GPU); N > 1 runs configs[2]'s per-GPU shape (batch 8 per GPU: 64 pairs
on 8 GPUs) unless --batch says otherwise.

The timed model is built with pipeline=True (tf_raft_amd/model.py "pipelined forward"): consecutive calls are independent, call n's
loop runs on lane n % 3 (a stream of its own) while the next calls' encoders, volume builds and loops proceed -- the timed region is
still K complete model([a, b]) calls between two device synchronisations.  RAFT_PIPELINE=0 times the serial schedule.

Rank 0 prints ONE JSON line; besides the contract fields it carries
  schedule            what was timed: loops in flight, launch-shape hint, the same K steps on the serial schedule, the latency of ONE call
                      consumed at once on both schedules
  final_iter_epe      the EPE half of BASELINE's metric ON THE TIMED CONFIGURATION: max-abs EPE of flow_predictions[-1] (element 0 of the
                      timed batch, Keras-default weights) against the CPU oracle; an untrained RAFT is ill conditioned there, so
                      final_iter_epe_within_tolerance is false and final_iter_epe_default_horizon says how far 1e-3 holds.
                      final_iter_epe_conditioned = worst of the regimes where 1e-3 is provable (conditioned, jump0); `parity` holds all 24
                      iterations, horizon and locality per regime (default, conditioned, mid, jump0)
  pairs_per_s_by_regime  the same step timed in every regime (3 interleaved rounds, median): no data-dependent work
  roofline            the dominant kernel (by accumulated time): achieved = the FLOPs the kernel EXECUTES on the MFMA
                      pipe per launch / its HIP-event time; frac = achieved / 157.3 TF.  With several loops in flight the kernel is
                      launched in the shape of a larger batch (fewer, longer workgroups) and shares the chip with the other loops'
                      kernels, so it is measured with the chip filled the same way: `instances` concurrent launches on as many
                      streams, instances x FLOPs / the slowest stream's time per launch (`rocprof` = the rocprofv3 average duration of
                      the same microbenchmark, profiles/kernel_concurrent.json); `single_instance` = one launch alone, with `rocprof`
                      from the committed single-stream trace (profiles/kernel_durations.json), each with its staleness.  A layer on
                      a Winograd kernel executes fewer multiplies than the direct convolution it computes; the
                      direct-convolution figure is kept as algorithmic_tflops / frac_algorithmic (may exceed 1)
  roofline_lookup_convc1_fused, roofline_mask_upsample_fused   the kernels the product loop runs
  roofline_corr_lookup the HBM-bound stand-alone lookup the north star singles out, at 4 / 8 / 16 pairs, with its 0.60 target
  roofline_corr_build  the volume build: bound "mfma" (its GEMM) with the HBM write figure beside it
  traffic             HBM bytes per launch from the in-loop B=8 PMC passes of profiles/pmc_traffic.json (tools/pmc_traffic.sh),
                      scaled per pair to this run's batch; `evidence` says whether those passes were taken on the running sources
  stage_ms            per-kernel average milliseconds per launch (HIP events, instrumented replay)
  cpu_baseline        the CPU oracle (reference restatement, torch-CPU) timed on this box's host cores, thread count probed
  preflight           (N > 1) announced on stderr before timing: ranks seen, devices, libraft_hip.so mapped / not rebuilt per rank
"""
import argparse
import contextlib
import ctypes as C
import json
import os
import sys
import time

# N > 1: besides the caller's stream and the three loop lanes a rank drives a gather stream and RCCL's own -- six streams on HIP's four
# default hardware queues would put a collective (a kernel that waits for its peers) in front of a loop's launches.  Eight queues keep
# every stream on a queue of its own; a single-GPU run (four streams) is left on the default (measured identical with 4 / 8 / 16:
# profiles/r12b_lanes_ab_per_process.txt).  Must be in the environment before the HIP runtime initialises.
if int(os.environ.get('WORLD_SIZE', '1')) > 1:
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H, W, ITERS = 448, 512, 24
METRIC = 'image-pairs/sec at 448\u00d7512 iters_pred=24; final-iter EPE vs TF ref'    # BASELINE.json "metric", verbatim
EPE_TOL = 1e-3                     # BASELINE.json north_star: flow_predictions[-1] within 1e-3 max-abs EPE
# Weight regimes the parity half of the metric is evaluated on (tf_raft_amd.weights; tests/golden/make_conditioning.py):
#   default      Keras-default random weights = BASELINE configs[1] (what `value` is timed on).  Flow grows ~7 px per iteration,
#                taps cross the sampler's discontinuities and ANY two fp32 evaluations part ways after ~10 iterations (the oracle
#                in fp32 against itself in fp64 included): the final EPE is reported with its horizon and locality, not bounded
#   conditioned  contractive flow head: sub-pixel flow, no tap near a discontinuity -- the regime where the 1e-3 bound is provable
#   mid          multi-pixel drifting flow: mostly well conditioned (reported with horizon / locality)
#   jump0        conditioned + an integer drift of (+1, -1) px per iteration: every lookup window moves across integers and the
#                clamped borders at every level in every iteration, yet no tap comes near a discontinuity -- provable as well
REGIMES = ('default', 'conditioned', 'mid', 'jump0')
PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec peak (6.29 TB/s measured copy)

STAGES = ['corr_lookup', 'convc1', 'convc2', 'convf1', 'convf2', 'conv', 'gru_zr1', 'gru_q1', 'gru_zr2',
          'gru_q2', 'fh1_mask0', 'fh2', 'mask2', 'upsample_convex']


# Layers that run on a Winograd kernel execute fewer multiplies than the convolution they compute: F(2x2, 3x3) 16 per
# 36 (conv_wino.h), F(4x4, 3x3) 36 per 144 (conv_wino4.h), F(2, 5) 6 per 10 and F(4, 5) 8 per 20 (conv_wino1d.h).  roofline.achieved counts the FLOPs EXECUTED on
# the MFMA pipe (direct FLOPs / this factor); the direct-convolution figure is kept as algorithmic_tflops.
WINOGRAD_ALGORITHMS = {2.25: 'Winograd F(2x2,3x3)', 4.0: 'Winograd F(4x4,3x3)', 10.0 / 6.0: 'Winograd F(2,5)', 2.5: 'Winograd F(4,5)'}


def winograd_layers(pairs=4):
    """{stage name: direct MACs / executed MACs} of the stages that are on a Winograd kernel under the current
    RAFT_CONV_WINO / RAFT_CONV_WINO4 / RAFT_GRU_WINO / RAFT_GRU_WINO4 switches (defaults of csrc/conv.hip: F(2x2,3x3) mask
    13 = convc2 | conv | fh1_mask0; F(4x4,3x3) mask 8 = fh1_mask0 from 2 pairs per launch on, + 1 | 2 = convc2, convf2 from 3, + 4 = conv from 8 on; GRU masks 15:
    F(4, 5) where its bit is set, else F(2, 5))."""
    from tf_raft_amd import _ffi
    m3 = int(_ffi.get_option('RAFT_CONV_WINO') or 13)
    m44 = _ffi.get_option('RAFT_CONV_WINO4')
    m44 = int(m44) if m44 else (0 if pairs < 2 else 8 | (3 if pairs >= 3 else 0) | (4 if pairs >= 8 else 0))
    mg = int(_ffi.get_option('RAFT_GRU_WINO') or 15)
    mg4 = int(_ffi.get_option('RAFT_GRU_WINO4') or 15)
    on = {}
    for bit, name in ((1, 'convc2'), (2, 'convf2'), (4, 'conv'), (8, 'fh1_mask0')):
        if m3 & bit:
            on[name] = 2.25
        if m44 & bit:
            on[name] = 4.0
    for bit, name in ((1, 'gru_zr1'), (2, 'gru_q1'), (4, 'gru_zr2'), (8, 'gru_q2')):
        if mg4 & bit:
            on[name] = 2.5
        elif mg & bit:
            on[name] = 10.0 / 6.0
    return on


def wino4_launch_shape(layer, B, h, w, conc=1):
    """(workgroups, K split?) of an F(4x4,3x3) launch of the update block -- the SAME rule as the library applies
    (csrc/conv_wino4.hip raft_launch_conv_wino4 + the per-layer hints of csrc/conv.hip update_basic_impl): 8-row x 64-pixel x
    64-channel workgroups, or K-split 4-row ones when RAFT_WINO4_KS says so, else when the layer's hint says so (convc2:
    RAFT_CONVC2_KS; convf2: RAFT_CONVF2_KS, default eight-row workgroups once there are >= 56 of them), else while the
    eight-row grid would be < 128.  The split needs input channels in multiples of 32 (every layer here has them)."""
    from tf_raft_amd import _ffi
    opt = lambda name: int(_ffi.get_option(name) or 0)
    nb = {'convc2': 3, 'fh1_mask0': 8, 'conv': 2, 'convf2': 1}[layer]
    tiles8 = B * ((h + 7) // 8) * ((w + 63) // 64)
    grid1 = tiles8 * nb
    # conc = raft_set_thread_concurrency: loops sharing the chip count a grid that many times
    hint = {'convc2': opt('RAFT_CONVC2_KS'), 'convf2': opt('RAFT_CONVF2_KS') or (1 if tiles8 * conc >= 56 else 0)}.get(layer, 0)
    ks = opt('RAFT_WINO4_KS') or (hint if hint in (1, 2) else (2 if grid1 * conc < 128 else 1))
    if ks == 2:
        return B * ((h + 3) // 4) * ((w + 63) // 64) * nb, True
    return grid1, False


# F(4x4) layers of the update block: stage -> (input channels, output channels, field of raft_basic_update_weights)
W44_FIELDS = {'convc2': (256, 192, 'convc2_w44'), 'conv': (256, 126, 'conv_w44'), 'fh1_mask0': (128, 512, 'fh1_mask0_w44'), 'convf2': (128, 64, 'convf2_w44')}


def concurrent_wino4_us(model, _dev, _ffi, field, cin, cout, B, h, w, conc, reps=30):
    """One F(4x4,3x3) layer of the update block launched on `conc` streams at once (own input / output per stream, the model's
    packed weights), under the calling thread's launch-shape hint: microseconds per launch of the slowest stream."""
    wt = getattr(model.update_block.c, field)
    lib = _dev.lib()
    streams = [torch.cuda.Stream() for _ in range(conc)]
    xs = [torch.randn((B, h, w, cin), device='cuda').relu_() for _ in range(conc)]
    outs = [torch.empty((B, h, w, cout), device='cuda') for _ in range(conc)]

    def launch(i):
        _ffi.check(lib.raft_conv2d_winograd4_f32(_dev.ptr(xs[i]), cin, cin, None, 0, 0, wt.wp, wt.bias, B, h, w, wt.npad, cout, 1, 1.0,
                                                 _dev.ptr(outs[i]), cout, streams[i].cuda_stream), 'conv2d_winograd4')
    torch.cuda.synchronize()
    for _ in range(3):
        for i in range(conc):
            launch(i)
    torch.cuda.synchronize()
    e0 = [torch.cuda.Event(enable_timing=True) for _ in range(conc)]
    e1 = [torch.cuda.Event(enable_timing=True) for _ in range(conc)]
    for i in range(conc):
        e0[i].record(streams[i])
    for _ in range(reps):
        for i in range(conc):
            launch(i)
    for i in range(conc):
        e1[i].record(streams[i])
    torch.cuda.synchronize()
    return max(e0[i].elapsed_time(e1[i]) for i in range(conc)) / reps * 1e3


def stage_work(B, h, w):
    """Algorithmic work per launch: MACs of the real (unpadded) convolution, or HBM bytes."""
    M = B * h * w
    macs = {
        'convc1': 324 * 256, 'convc2': 9 * 256 * 192, 'convf1': 49 * 2 * 128, 'convf2': 9 * 128 * 64,
        # SepConvGRU rows of `inp` (128 of the 384 input channels) are loop-invariant and evaluated once per
        # forward by raft_gru_context_f32 (pre-loop 'gru_context': 2 x 5*128*384 MAC/px): executed K = 5*256
        'conv': 9 * 256 * 126, 'gru_zr1': 5 * 256 * 256, 'gru_q1': 5 * 256 * 128, 'gru_zr2': 5 * 256 * 256,
        'gru_q2': 5 * 256 * 128, 'fh1_mask0': 9 * 128 * 512, 'fh2': 9 * 256 * 2, 'mask2': 256 * 576,
    }
    flops = {k: 2.0 * v * M for k, v in macs.items()}
    bytes_ = {
        # 4 levels x (2r+2)^2 footprint reads + coords, 324 outputs written
        'corr_lookup': M * (4 * 100 * 4 + 8 + 324 * 4),
        # mask 576 + flow 2 read, 8x8x2 written
        'upsample_convex': M * (576 * 4 + 8 + 64 * 2 * 4),
    }
    return flops, bytes_


def measured_copy_gbs(device, lib, _dev, check, floats=1 << 28, reps=10):
    """HBM copy bandwidth of THIS box (read + write bytes / time) with the library's 16-byte-per-lane streaming
    copy: 2 x 1 GiB buffers (4x the 256 MiB Infinity Cache), HIP events on the launch stream."""
    src = torch.empty(floats, device=device, dtype=torch.float32).normal_()
    dst = torch.empty_like(src)
    run = lambda: check(lib.raft_stream_copy_f32(_dev.ptr(src), _dev.ptr(dst), floats, _dev.stream_ptr()), 'stream_copy')
    for _ in range(2):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * 4.0 * floats * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def measured_mfma_tflops(device, lib, _dev, check):
    """fp32-MFMA rate THIS box sustains (raft_mfma_probe_f32: nothing but v_mfma_f32_16x16x4_f32, 2 waves per SIMD on every
    CU, non-zero operands), HIP events on the launch stream: `short` = one ~1.5 ms launch from an idle chip, `sustained` = the
    last 20 of 60 back-to-back launches (~100 ms of continuous fp32 MFMA: the power-limited clock the prediction loop runs at)."""
    blocks, iters = 512, 8192
    out = torch.empty(blocks * 256, device=device, dtype=torch.float32)
    flop = blocks * 4 * iters * 8 * 2048.0
    run = lambda: check(lib.raft_mfma_probe_f32(_dev.ptr(out), blocks, iters, _dev.stream_ptr()), 'mfma_probe')
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    run()                                   # module load
    torch.cuda.synchronize()
    time.sleep(0.2)                         # idle chip
    ev[0].record(); run(); ev[1].record()
    torch.cuda.synchronize()
    short = flop / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e12
    for _ in range(40):
        run()
    ev[2].record()
    for _ in range(20):
        run()
    ev[3].record()
    torch.cuda.synchronize()
    sustained = 20 * flop / (ev[2].elapsed_time(ev[3]) * 1e-3) / 1e12
    return short, sustained


def pmc_traffic(kernel, B):
    """(HBM bytes per launch at batch B, source note) of `kernel` from the committed rocprofv3 --pmc passes
    (profiles/pmc_traffic.json, regenerated by tools/pmc_traffic.sh + tools/pmc_traffic.py: FETCH_SIZE and WRITE_SIZE in
    separate passes over the single-stream loop at B = 8 -- a 550 MB volume, larger than the 256 MiB Infinity Cache --
    FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM: gfx950 tallies 128-byte requests at 64 bytes).  The pass's
    per-launch bytes are scaled per pair to this run's batch.  The note carries `stale: true` when the passes were taken on
    other kernel sources than the ones this process runs (evidence_file).  (None, None) when no pass covers the kernel."""
    # passes taken AT this run's batch (profiles/pmc_traffic_b<B>.json, tools/closing_set.sh) are preferred: no scaling, and
    # the volume's relation to the 256 MiB Infinity Cache is the timed one (275 MB at 4 pairs straddles it)
    d, ev = evidence_file(f'pmc_traffic_b{B}.json')
    t = (d or {}).get(kernel)
    if not t or not t.get('batch') or ev.get('stale'):
        d, ev = evidence_file('pmc_traffic.json')
        t = (d or {}).get(kernel)
    if not t or not t.get('batch'):
        return None, None
    per_pair = t['hbm_bytes_per_launch'] / t['batch']
    note = {'source': t.get('source'), 'pass_batch': t['batch'], 'in_loop': bool(t.get('in_loop')),
            'hbm_bytes_per_pair': round(per_pair), 'algorithmic_bytes_per_pair': t.get('algorithmic_bytes_per_pair'),
            'scaled_to_batch': B, 'stale': ev['stale'], 'measured_at_commit': ev.get('measured_at_commit')}
    return int(round(per_pair * B)), note


def evidence_file(name):
    """A committed evidence summary under profiles/ (pmc_traffic.json, kernel_durations.json) and whether it was taken from
    THIS build: the file records the digest of the HIP sources + flags it was measured on (tf_raft_amd.build.source_digest());
    a mismatch is reported as stale (the numbers then describe older kernels)."""
    from tf_raft_amd import build as _build
    try:
        with open(os.path.join(ROOT, 'profiles', name)) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None, {'file': f'profiles/{name}', 'present': False, 'stale': True}
    meta = d.get('_meta', {})
    same = bool(meta.get('source_digest')) and meta.get('source_digest') == _build.source_digest()
    return d, {'file': f'profiles/{name}', 'present': True, 'stale': not same, 'measured_at_commit': meta.get('git_head'),
               'source_digest': (meta.get('source_digest') or '')[:16], 'build_digest': _build.source_digest()[:16],
               'raw_files': meta.get('raw_files')}


def rocprof_us(durations, batch, key):
    """Average rocprofv3 kernel duration (us) of stage `key` at `batch` pairs from profiles/kernel_durations.json
    (tools/kernel_durations.py over the single-stream loop of tools/pmc_loop.py), None when not covered."""
    try:
        return float(durations[f'b{batch}'][key]['avg_us'])
    except (KeyError, TypeError, ValueError):
        return None


def parity_stats(got, want, tol=EPE_TOL):
    """flow_predictions of ONE pair (lists of (1,H,W,2) arrays: HIP, oracle) -> final-iteration max-abs EPE, the per-iteration
    series, the horizon (first iteration beyond `tol`) and, when there is a departure, how local it is."""
    per_px = [np.sqrt(((np.asarray(g, dtype=np.float64) - np.asarray(w_, dtype=np.float64)) ** 2).sum(-1)) for g, w_ in zip(got, want)]
    errs = [float(d.max()) for d in per_px]
    first = next((i for i, e in enumerate(errs) if e > tol), len(errs))
    out = {'final_iter_epe': float(f'{errs[-1]:.3e}'), 'per_iteration_epe': [float(f'{e:.2e}') for e in errs],
           'iterations_within_tol': first, 'within_tol_on_every_iteration': first == len(errs),
           'max_abs_flow_px': round(float(np.abs(want[-1]).max()), 2)}
    if first < len(errs):
        d = per_px[first]
        out['first_departure'] = {'iteration': first, 'frac_pixels_within_tol': round(float((d <= tol).mean()), 4),
                                  'median_pixel_epe': float(f'{float(np.median(d)):.2e}')}
        out['final_frac_pixels_within_tol'] = round(float((per_px[-1] <= tol).mean()), 4)
    return out


def best_oracle_threads(run, candidates):
    """The CPU restatement is an eager torch graph of small ops: more threads than it can use make it slower (128 threads:
    6 s per pair on the GPU boxes, 8 threads: 2.6 s in the build container).  Time a 2-iteration forward per candidate and
    keep the fastest -- the thread count is what cpu_baseline.cores reports."""
    best, best_t = None, None
    for n in candidates:
        torch.set_num_threads(n)
        run()                                   # thread-pool warm-up at this size
        t0 = time.perf_counter()
        run()
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best, best_t = n, dt
    torch.set_num_threads(best)
    return best


def preflight(dist, world, rank, local_rank, device, backend):
    """Before anything is timed with N > 1 ranks: every rank reports its device, whether libraft_hip.so is mapped into the
    process and whether this process had to COMPILE it (a box that received the prebuilt .so must not), through an
    all_gather_object over the job's backend; rank 0 prints the table to stderr and returns the summary for the line."""
    import socket
    from tf_raft_amd import _ffi, build as _build
    lib = _ffi.load_library()
    with open('/proc/self/maps') as f:
        mapped = 'libraft_hip.so' in f.read()
    mine = {'rank': rank, 'local_rank': local_rank, 'pid': os.getpid(), 'host': socket.gethostname(),
            'device_index': device.index, 'device_name': torch.cuda.get_device_name(device),
            'so_mapped': mapped, 'so_compiled_by_this_rank': bool(_build.LAST_BUILD_COMPILED), 'abi': int(lib.raft_version()),
            'build_digest': _build.source_digest()[:16]}
    table = [None] * world
    dist.all_gather_object(table, mine)
    seen = ranks_seen(dist, world, rank, device)
    summary = {'ranks_seen': seen, 'world_size': world,
               'backend': ('RCCL ' + '.'.join(str(v) for v in torch.cuda.nccl.version())) if backend == 'nccl' else backend,
               'device_index_by_rank': [t['device_index'] for t in table],
               'distinct_devices': len({(t['host'], t['device_index']) for t in table}),
               'so_mapped_on_every_rank': all(t['so_mapped'] for t in table),
               'ranks_that_compiled_the_so': [t['rank'] for t in table if t['so_compiled_by_this_rank']],
               'same_build_on_every_rank': len({t['build_digest'] for t in table}) == 1}
    if rank == 0:
        print('[bench preflight] ' + json.dumps({'summary': summary, 'ranks': table}), file=sys.stderr, flush=True)
    return summary


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-run this command line as N ranks (one per GPU) through
    torch.distributed.run, rendezvous on 127.0.0.1 at a free port.  Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL needs it on this driver
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def ranks_seen(dist, world, rank, device):
    """Every rank contributes its id to an all-gather over the job's backend; the number of distinct ids that arrive
    is what the line reports as `ranks_seen` (world size when the collective really spans all ranks)."""
    import torch
    mine = torch.tensor([rank], device=device, dtype=torch.int64)
    out = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return len({int(t.item()) for t in out})


def dry_run_cpu(args, world, rank):
    """RAFT_BENCH_DRY_RUN=cpu: the launcher / rendezvous / barrier / in-flight gather / max-over-ranks / rank-0 line
    control flow of `--gpus N` with NO model and NO GPU (gloo, a zero tensor of the prediction's shape per step).
    Exists so that tests/test_distributed_cpu.py can run `python bench.py --gpus 2` end to end in a CPU container;
    the line says `dry_run` and its value means nothing."""
    import torch
    import torch.distributed as dist
    from tf_raft_amd.parallel import all_gather_batch_async
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    B = args.batch or 2
    pending = []
    if world > 1:       # the same announcement the GPU path makes before timing (no library and no device in a dry run)
        table = [None] * world
        dist.all_gather_object(table, {'rank': rank, 'pid': os.getpid(), 'local_rank': int(os.environ.get('LOCAL_RANK', '0'))})
        if rank == 0:
            print('[bench preflight] ' + json.dumps({'summary': {'ranks_seen': len({t['rank'] for t in table}), 'world_size': world,
                                                                'backend': 'gloo (dry run)'}, 'ranks': table}), file=sys.stderr, flush=True)

    def step():
        last = torch.full((B, 8, 8, 2), float(rank))
        if world > 1:
            pending.append(all_gather_batch_async(last, world * B))
            if len(pending) > 1:
                return pending.pop(0).wait()
        return last

    def fence():
        while pending:
            got = pending.pop(0).wait()
            assert got.shape[0] == world * B and float(got[-1, 0, 0, 0]) == world - 1
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    seen = 1
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        seen = ranks_seen(dist, world, rank, torch.device('cpu'))
    if rank == 0:
        print(json.dumps({
            'metric': METRIC, 'value': 0.0, 'unit': 'image-pairs/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'none',
            'dry_run': 'control flow only: no model, no GPU (RAFT_BENCH_DRY_RUN=cpu)', 'ranks_seen': seen,
            'config': {'workload': 'DRY RUN', 'pairs_per_gpu': B, 'global_batch': world * B, 'parallelism': f'dp{world}',
                       'collective': 'all_gather over gloo (dry run)'}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def train_bench(args, world, rank, device, backend):
    """`--train`: RAFT.train_step (forward with tape, sequence_loss, backward through time, clip, AdamW) on synthetic data at
    the reference's training crop (configs/train_chairs.yml: 368x496, iters 12), batch 4 per GPU; with N > 1 ranks every step
    all-reduces the 5.26 M gradients (one bucket) and the batch-norm batch statistics over the job's backend.  Reported next
    to the inference headline, never instead of it."""
    import torch.distributed as dist
    import tf_raft_amd
    from tf_raft_amd import losses, training
    from tf_raft_amd import weights as wm
    B, Ht, Wt, iters = args.batch or 4, 368, 496, 12
    rng = np.random.default_rng(100 + rank)
    model = tf_raft_amd.RAFT(weights=wm.init_weights('raft', seed=0), iters=iters, iters_pred=24)
    sched = training.CyclicalLearningRate(4e-4, 8e-4, 1000, training.first_cycle_scaler)
    model.compile(optimizer=training.AdamW(1e-4, sched), clip_norm=1.0, loss=losses.sequence_loss, epe=losses.end_point_error,
                  tape_dtype=args.tape)
    gen = torch.Generator(device=device)
    gen.manual_seed(2000 + rank)
    data = (torch.rand((B, Ht, Wt, 3), device=device, generator=gen) * 255.0, torch.rand((B, Ht, Wt, 3), device=device, generator=gen) * 255.0,
            torch.randn((B, Ht, Wt, 2), device=device, generator=gen) * 3.0, torch.ones((B, Ht, Wt), device=device, dtype=torch.bool))

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(max(1, args.warmup)):
        res = model.train_step(data)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = model.train_step(data)
    fence()
    elapsed = time.perf_counter() - t0
    seen = 1
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        seen = ranks_seen(dist, world, rank, device)
    if rank == 0:
        print(json.dumps({
            'metric': 'training image-pairs/sec at 368x496 iters=12 (RAFT.train_step)', 'value': round(world * B * args.steps / elapsed, 3),
            'unit': 'image-pairs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': max(1, args.warmup),
            'ms_per_step': round(1e3 * elapsed / args.steps, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if args.tape == 'f32' else 'f32 arithmetic, bf16 tape storage', 'data': 'synthetic', 'ranks_seen': seen,
            'loss_last_step': float(res['loss']), 'peak_mem_gib': round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
            'config': {'workload': f'RAFT train_step, BASELINE configs[4] per-GPU shape: batch {B}/GPU {Ht}x{Wt} iters={iters}, AdamW + '
                                   f'clip_by_global_norm, tape {args.tape}',
                       'pairs_per_gpu': B, 'global_batch': world * B, 'parallelism': f'dp{world}',
                       'collective': ('all_reduce(gradients, 21 MB in one bucket) + all_reduce(batch-norm statistics) over '
                                      + ('RCCL' if backend == 'nccl' else backend)) if world > 1 else 'none'}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)     # the driver's values; with several loops in flight a short timed region is
    ap.add_argument('--warmup', type=int, default=5)     # mostly ramp and drain (5 steps: 324 pairs/s, 20: 377, 80: 385)
    ap.add_argument('--batch', type=int, default=None,
                    help='image pairs per GPU per step (default: 4 on one GPU = BASELINE configs[1]; 8 per GPU on N > 1 '
                         'GPUs = configs[2], 64 pairs on 8 GPUs)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--train', action='store_true',
                    help='time RAFT.train_step instead (BASELINE configs[4] per-GPU shape: batch 4, 368x496, iters 12; gradients '
                         'all-reduced over the job backend when --gpus N > 1); NOT the headline metric')
    ap.add_argument('--tape', default='f32', choices=['f32', 'bf16'], help='--train: storage type of the activation tape')
    ap.add_argument('--cpu-runs', type=int, default=4, help='minimum number of timed CPU-oracle forwards (one per weight regime first)')
    ap.add_argument('--no-parity', action='store_true', help='skip the per-regime timing + EPE legs (profiling runs)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher -- one rank per GPU under torch.distributed.run on
        # 127.0.0.1 (the container hostname may not resolve); rank 0's JSON line is this process's output
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU')
    if os.environ.get('RAFT_BENCH_DRY_RUN') == 'cpu':
        return dry_run_cpu(args, world, rank)
    import torch.distributed as dist
    # RAFT_BENCH_BACKEND=gloo: dry run of the N > 1 control flow on a box with fewer GPUs than ranks (ranks then share
    # devices round-robin; gloo stages the device tensors through the host) -- never a measurement
    backend = os.environ.get('RAFT_BENCH_BACKEND', 'nccl')
    dev_index = local_rank if backend == 'nccl' else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    pre = preflight(dist, world, rank, local_rank, device, backend) if world > 1 else None
    if args.train:
        return train_bench(args, world, rank, device, backend)

    import tf_raft_amd
    from tf_raft_amd import _dev, _ffi
    from tf_raft_amd import weights as wm
    from tf_raft_amd.layers.corr import CorrBlock
    from tf_raft_amd.parallel import all_gather_batch, all_gather_batch_async

    B = args.batch if args.batch else (4 if world == 1 else 8)
    cfg_name = 'BASELINE configs[1]' if (world == 1 and B == 4) else (
        'BASELINE configs[2] per-GPU shape' if B == 8 else 'custom batch')
    wts = wm.init_weights('raft', seed=0)
    # The timed steps are back-to-back calls whose results nobody consumes at once: the pipelined schedule (opt-in since round 6,
    # ADVICE r5) with its default number of loops in flight.  RAFT_PIPELINE=0 times the serial schedule instead.
    pipelined = os.environ.get('RAFT_PIPELINE', '1') != '0'
    model = tf_raft_amd.RAFT(iters_pred=ITERS, weights=wts, pipeline=pipelined)
    conc = model.lanes if (pipelined and model.lanes > 1 and model._shape_hint != 'none') else 1      # the launch-shape hint the product's loops run under
    gen = torch.Generator(device=device)
    gen.manual_seed(1000 + rank)
    img1 = torch.rand((B, H, W, 3), device=device, generator=gen) * 255.0
    img2 = torch.rand((B, H, W, 3), device=device, generator=gen) * 255.0

    pending = []          # N > 1: the all-gather of step i's final predictions is in flight while step i + 1 computes
    gather_async = [os.environ.get('RAFT_BENCH_BLOCKING_GATHER', '0') != '1']

    # N > 1: the gather is issued from a stream of its own.  The model's loop runs on the library's 'loop' stream and the compute
    # stream is NOT made to wait for it (pipelined forward: the next step's encoders run under this step's loop); touching
    # preds[-1] joins the touching stream with the loop, so it is touched on `comm`, and RCCL orders its own stream behind `comm`.
    comm = torch.cuda.Stream(device=device) if (world > 1 and device.type == 'cuda') else None

    def step(a=img1, b=img2):
        preds = model([a, b], training=False)
        if world > 1:
            with (torch.cuda.stream(comm) if comm is not None else contextlib.nullcontext()):
                last = preds[-1].as_subclass(torch.Tensor)
                if comm is not None:
                    last.record_stream(comm)
                if gather_async[0]:
                    try:
                        pending.append(all_gather_batch_async(last, world * B))
                    except (RuntimeError, TypeError, NotImplementedError) as e:   # backend without async collectives
                        print(f'[bench] async all-gather unavailable ({e}); using the blocking gather', file=sys.stderr)
                        gather_async[0] = False
                if not gather_async[0]:
                    return all_gather_batch(last, world * B)
                if len(pending) > 1:
                    return pending.pop(0).wait()
                return None
        return preds[-1]

    def fence():
        while pending:
            pending.pop(0).wait()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    rank_elapsed = [elapsed]
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        rank_elapsed = [float(e.item()) for e in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    value = world * B * args.steps / elapsed
    seen = ranks_seen(dist, world, rank, device) if world > 1 else 1

    # ---------------- the same step in every weight regime: throughput must not depend on the weights (no data-dependent work
    # is skipped when the flow runs off the map), and element 0's 24 predictions are kept for the EPE half of the metric
    regime_weights, regime_pred0, regime_rate, regime_rounds = {}, {}, {}, {}
    if world == 1 and not args.no_parity:
        models = {}
        for reg in REGIMES:
            regime_weights[reg] = wts if reg == 'default' else wm.condition_weights('raft', wts, reg)
            models[reg] = model if reg == 'default' else tf_raft_amd.RAFT(iters_pred=ITERS, weights=regime_weights[reg], pipeline=pipelined)
            for _ in range(max(1, args.warmup)):
                models[reg]([img1, img2], training=False)
        torch.cuda.synchronize()
        rounds = 3                      # interleaved: box drift and allocator effects hit every regime alike; median per regime
        for _ in range(rounds):
            for reg in REGIMES:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    models[reg]([img1, img2], training=False)
                torch.cuda.synchronize()
                regime_rounds.setdefault(reg, []).append(B * args.steps / (time.perf_counter() - t0))
        for reg in REGIMES:
            regime_rate[reg] = round(float(np.median(regime_rounds[reg])), 3)
            preds = models[reg]([img1, img2], training=False)
            torch.cuda.synchronize()
            regime_pred0[reg] = [p_.as_subclass(torch.Tensor)[:1].cpu().numpy() for p_ in preds]
            del preds
        del models
    elif world > 1 and rank == 0 and not args.no_parity:
        # N > 1: rank 0 alone evaluates the two provable regimes on ITS shard's element 0 (the other ranks wait at the closing
        # barrier); the CPU oracle runs below, untimed -- cpu_baseline is an N = 1 figure
        for reg in ('conditioned', 'jump0'):
            regime_weights[reg] = wm.condition_weights('raft', wts, reg)
            preds = tf_raft_amd.RAFT(iters_pred=ITERS, weights=regime_weights[reg], pipeline=pipelined)([img1, img2], training=False)
            torch.cuda.synchronize()
            regime_pred0[reg] = [p_.as_subclass(torch.Tensor)[:1].cpu().numpy() for p_ in preds]
            del preds

    result = {
        'metric': METRIC, 'value': round(value, 3), 'unit': 'image-pairs/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(1e3 * elapsed / args.steps, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'RAFT forward {cfg_name}: batch {B}/GPU {H}x{W} iters_pred={ITERS} random weights',
                   'detail': f'all {ITERS} upsampled predictions produced per pair, Keras-default random weights, inputs '
                             'resident in HBM',
                   'pairs_per_gpu': B, 'global_batch': world * B, 'parallelism': f'dp{world}',
                   'collective': ('all_gather(flow_predictions[-1]) over ' + ('RCCL' if backend == 'nccl' else backend + ' (DRY RUN of the control flow, not a measurement)')
                                  + (', in flight under the next step' if gather_async[0] else '')) if world > 1 else 'none'},
    }

    result['schedule'] = {
        'pipeline': bool(pipelined), 'loops_in_flight': model.lanes if pipelined else 1,
        'loop_streams': ('one stream per loop (single-stream schedule), one lane per loop in flight' if model.lanes > 1 else
                         'three streams per loop (chain + flow branch + mask branch)') if pipelined else 'three streams per loop, calls serial',
        'launch_shape_hint': conc, 'launch_shape_hint_scope': model._shape_hint if conc > 1 else 'none',
        'hip_hardware_queues': os.environ.get('GPU_MAX_HW_QUEUES', 'default (4)'),
        'note': 'the timed region is K back-to-back model([a, b]) calls between two device synchronisations: every kernel of every call is '
                'inside it.  Consecutive calls are independent; call n runs its loop on lane n % loops_in_flight while the next calls\' '
                'encoders, volume builds and loops proceed (tf_raft_amd/model.py "pipelined forward"); per-call results equal the serial '
                'schedule\'s bit for bit under the same launch-shape hint (tests/test_gpu_model.py::test_pipelined_calls_are_bitwise_the_serial_calls)'}
    if world == 1 and rank == 0:
        # the same K steps on the serial schedule (RAFT(pipeline=False): the default of a model object, what a caller that consumes
        # every result at once gets), and the latency of ONE call whose result is consumed at once, on both schedules
        serial_model = tf_raft_amd.RAFT(iters_pred=ITERS, weights=wts, pipeline=False)
        for _ in range(max(2, args.warmup)):
            serial_model([img1, img2], training=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            serial_model([img1, img2], training=False)
        torch.cuda.synchronize()
        result['schedule']['serial_schedule_pairs_per_s'] = round(B * args.steps / (time.perf_counter() - t0), 3)

        def consumed_ms(m, a, b, n=5):
            ts = []
            for _ in range(n):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out = m([a, b], training=False)[-1].as_subclass(torch.Tensor)
                torch.cuda.current_stream().synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
                del out
            return round(float(np.median(ts)), 3)
        result['schedule']['single_call_consumed_at_once_ms'] = {
            'pipelined_model': consumed_ms(model, img1, img2), 'serial_model': consumed_ms(serial_model, img1, img2),
            'one_pair_pipelined_model': consumed_ms(model, img1[:1], img2[:1]), 'one_pair_serial_model': consumed_ms(serial_model, img1[:1], img2[:1])}
        del serial_model

    if world > 1:
        result['ranks_seen'] = seen                      # distinct rank ids that arrived through the job's all-gather
        result['preflight'] = pre                        # announced on stderr BEFORE the timed region (devices, .so mapped / not rebuilt)
        result['scaling_curve'] = 'none measured by this repository: 8-GPU runs are the driver\'s (DESIGN.md section 6)'
        result['backend'] = 'RCCL ' + '.'.join(str(v) for v in torch.cuda.nccl.version()) if backend == 'nccl' else backend
        # the collective alone: one all-gather of flow_predictions[-1] (every rank takes part), barrier-to-sync, median of 5
        last = model([img1, img2], training=False)[-1].as_subclass(torch.Tensor)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            all_gather_batch(last, world * B)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e6)
        result['all_gather_us'] = {'median': round(float(np.median(ts)), 1), 'bytes_per_rank': int(last.numel() * 4),
                                   'note': 'issued alone after a barrier; in the timed steps it is in flight under the next step'}
        del last
        # ONE GPU at the same per-GPU shape, no collective: rank 0 alone, the other ranks wait at the closing barrier.
        # The driver computes scaling efficiency from its own N = 1 run; this is the like-for-like figure beside it.
        if rank == 0:
            for _ in range(2):
                model([img1, img2], training=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n1 = max(1, args.steps)
            for _ in range(n1):
                model([img1, img2], training=False)
            torch.cuda.synchronize()
            result['one_gpu_same_shape_pairs_per_s'] = round(B * n1 / (time.perf_counter() - t0), 3)
            # efficiency against the SAME per-GPU shape on this box (not against the 4-pair N = 1 line, whose
            # smaller batch fills the chip less well); the driver computes its own figure from its own N = 1 run
            result['scaling_efficiency'] = round(value / world / result['one_gpu_same_shape_pairs_per_s'], 4)
            result['scaling_efficiency_basis'] = ('whole-job pairs/s / n_gpus / one_gpu_same_shape_pairs_per_s (rank 0 alone, the same '
                                                  f'{B} pairs per step, no collective, measured in this job after the timed region)')
        result['rank_step_ms'] = {'min': round(1e3 * min(rank_elapsed) / args.steps, 3), 'max': round(1e3 * max(rank_elapsed) / args.steps, 3),
                                  'per_rank': [round(1e3 * e / args.steps, 3) for e in rank_elapsed]}

    if rank == 0 and world == 1:
        # ---------------- informational: RAFT.predict_step (flow_predictions[-1] only: mask head + upsampling in the
        # last iteration only, reference model.py:160-166) -- NOT the headline, which produces all 24 predictions
        for _ in range(2):
            model.predict_step((img1, img2))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = max(1, args.steps)                      # several loops are in flight: as many steps as the headline, or the ramp dominates
        for _ in range(n):
            model.predict_step((img1, img2))
        torch.cuda.synchronize()
        result['predict_step_pairs_per_s'] = round(B * n / (time.perf_counter() - t0), 3)
        # ---------------- informational: the reference's canonical call shape (README.md:98-103), ONE (1,448,512,3) pair
        for _ in range(2):
            model([img1[:1], img2[:1]], training=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            model([img1[:1], img2[:1]], training=False)
        torch.cuda.synchronize()
        # back-to-back calls, results not consumed: a THROUGHPUT figure (the latency of one consumed call is schedule.single_call_consumed_at_once_ms)
        result['batch1_back_to_back_ms_per_call'] = round(1e3 * (time.perf_counter() - t0) / n, 3)
        result['batch1_pairs_per_s'] = round(n / (time.perf_counter() - t0), 3)
        # ---------------- informational: ONE GPU at the per-GPU shape `--gpus N > 1` runs (BASELINE configs[2]: 8 pairs
        # per GPU), so that multi-GPU lines can be set against a single-GPU number of the same per-GPU work
        if B != 8:
            i8a = torch.cat([img1, img1.flip(0)])[:8] if B >= 4 else img1[:1].expand(8, -1, -1, -1).contiguous()
            i8b = torch.cat([img2, img2.flip(0)])[:8] if B >= 4 else img2[:1].expand(8, -1, -1, -1).contiguous()
            for _ in range(2):
                model([i8a, i8b], training=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                model([i8a, i8b], training=False)
            torch.cuda.synchronize()
            result['pairs_per_s_at_8_pairs_per_gpu'] = round(8 * n / (time.perf_counter() - t0), 3)
            del i8a, i8b

    if rank == 0:
        # ---------------- instrumented replay: per-kernel HIP-event timing on the launch stream, with the launch shapes the
        # product's loops run (the thread-local hint of the multi-lane schedule; reset before the line is printed)
        _dev.lib().raft_set_thread_concurrency(conc)
        h, w = H // 8, W // 8
        x1 = 2 * (img1 / 255.0) - 1.0
        x2 = 2 * (img2 / 255.0) - 1.0
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        torch.cuda.synchronize()
        ev[0].record()
        fmap1, fmap2 = model.fnet([x1, x2])
        ev[1].record()
        cnet = model.cnet(x1)
        ev[2].record()
        corr = CorrBlock(fmap1, fmap2, num_levels=4, radius=4)
        ev[3].record()
        st = model._get_state(B, h, w, device)
        model._prepare(cnet, st)
        ev[4].record()
        torch.cuda.synchronize()
        pre_ms = {'fnet': ev[0].elapsed_time(ev[1]), 'cnet': ev[1].elapsed_time(ev[2]),
                  'corr_build': ev[2].elapsed_time(ev[3]), 'prepare_state+gru_context': ev[3].elapsed_time(ev[4])}
        flow_up = torch.empty((ITERS, B, H, W, 2), device=device)
        reps = max(1, min(args.steps, 5))

    def timed_replay():
        acc = np.zeros(len(STAGES), dtype=np.float64)
        buf = (C.c_float * len(STAGES))()
        for _ in range(reps):
            model._prepare(cnet, st)
            _ffi.check(_dev.lib().raft_iterate_basic_timed_f32(
                C.byref(model.update_block.c), _dev.ptr(corr._pyr), corr._off, B, h, w, ITERS, C.byref(st.c),
                _dev.ptr(flow_up), _dev.stream_ptr(), buf), 'iterate_basic_timed')
            acc += np.array(list(buf), dtype=np.float64)
        per_launch_ms = acc / (reps * ITERS)
        return per_launch_ms

    if rank == 0:
        # per-kernel numbers come from the TWO-kernel path (stand-alone lookup, stand-alone convc1: the kernels the
        # roofline entries name); the product loop runs them fused (RAFT_LOOKUP_FUSED, default on), timed separately
        _ffi.set_option('RAFT_LOOKUP_FUSED', 0)
        _ffi.set_option('RAFT_MASK_FUSED', 0)                 # likewise mask.2 and the upsampling (RAFT_MASK_FUSED, default on)
        per_launch_ms = timed_replay()
        # What an event bracket adds to every stage: the SAME single-stream launches without events in between
        # (raft_iterate_basic_f32), timed with one event pair, against the sum of the 14 bracketed stages.  An event record
        # between two kernels keeps the second one from being dispatched under the first one's tail; rocprofv3's kernel
        # durations of the single-stream loop (profiles/) agree with the stage times once this is taken off.
        plain = []
        for _ in range(reps + 1):
            model._prepare(cnet, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _ffi.check(_dev.lib().raft_iterate_basic_f32(
                C.byref(model.update_block.c), _dev.ptr(corr._pyr), corr._off, B, h, w, ITERS, C.byref(st.c),
                _dev.ptr(flow_up), _dev.stream_ptr()), 'iterate_basic')
            e1.record()
            torch.cuda.synchronize()
            plain.append(e0.elapsed_time(e1) / ITERS)
        plain_iter_ms = float(np.median(plain[1:]))
        bracket_ms = max(0.0, (float(per_launch_ms.sum()) - plain_iter_ms) / len(STAGES))
        _ffi.set_option('RAFT_LOOKUP_FUSED', None)
        _ffi.set_option('RAFT_MASK_FUSED', None)
        fused_ms = timed_replay()
        stage_ms_events = {k: round(float(v), 5) for k, v in zip(STAGES, per_launch_ms)}
        stage_ms = {k: round(max(float(v) - bracket_ms, 1e-6), 5) for k, v in zip(STAGES, per_launch_ms)}
        acc = per_launch_ms
        flops, bytes_ = stage_work(B, h, w)
        dom = STAGES[int(np.argmax(acc))]
        # convc2 and conv are both F(4x4) launches of ~100 us in the lane shapes and trade places from run to run: within 3 % of the
        # longest stage the line stays on convc2 -- the kernel every round's line, VERDICT and the committed evidence files name
        if acc[STAGES.index('convc2')] >= 0.97 * float(acc.max()):
            dom = 'convc2'
        # lookup stage (empty bracket) + convc1 stage (the fused kernel): two brackets around one kernel
        fused_us = max(float(fused_ms[0] + fused_ms[1] - 2 * bracket_ms), 1e-3) * 1e3
        copy_gbs = measured_copy_gbs(device, _dev.lib(), _dev, _ffi.check)
        result['hbm_copy_gbs_measured'] = round(copy_gbs, 1)
        mfma_short, mfma_sust = measured_mfma_tflops(device, _dev.lib(), _dev, _ffi.check)
        result['mfma_fp32_tflops_measured'] = {
            'short_launch_from_idle': round(mfma_short, 1), 'sustained': round(mfma_sust, 1), 'spec': PEAK_FP32_MFMA_TFLOPS,
            'sustained_over_spec': round(mfma_sust / PEAK_FP32_MFMA_TFLOPS, 4),
            'note': 'raft_mfma_probe_f32: an MFMA-only loop (no loads, no VALU) on all 256 CUs; every roofline.frac is quoted against '
                    'the 157.3 TFLOP/s datasheet peak as the contract asks, frac_of_sustained_mfma beside it is against this figure'}

        wl = winograd_layers(B * conc)                        # the library's rules count a grid `conc` times under the hint
        if dom in flops:
            ratio = wl.get(dom, 1.0)                          # direct MACs / MACs issued on the MFMA pipe
            alg = flops[dom] / (stage_ms[dom] * 1e-3) / 1e12
            ach = alg / ratio
            tr, note = pmc_traffic(dom, B)
            roof = {'kernel': dom, 'bound': 'mfma', 'achieved': round(ach, 2), 'peak': PEAK_FP32_MFMA_TFLOPS,
                    'unit': 'TFLOP/s', 'frac': round(ach / PEAK_FP32_MFMA_TFLOPS, 4), 'traffic': tr,
                    'flops_per_launch': flops[dom] / ratio, 'ms_per_launch': stage_ms[dom],
                    'ms_per_launch_between_events': stage_ms_events[dom], 'traffic_source': note,
                    'flops_counted': 'executed on the MFMA pipe',
                    'achieved_between_events': round(flops[dom] / ratio / (stage_ms_events[dom] * 1e-3) / 1e12, 2),
                    'frac_between_events': round(flops[dom] / ratio / (stage_ms_events[dom] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                    'frac_of_sustained_mfma': round(ach / mfma_sust, 4)}
            if ratio != 1.0:
                roof['algorithm'] = WINOGRAD_ALGORITHMS[ratio]
                roof['direct_conv_flops_per_launch'] = flops[dom]
                roof['algorithmic_tflops'] = round(alg, 2)    # direct-convolution FLOPs / time: may exceed the peak
                roof['frac_algorithmic'] = round(alg / PEAK_FP32_MFMA_TFLOPS, 4)
            if ratio == 4.0 and dom in ('convc2', 'fh1_mask0', 'conv', 'convf2'):
                # launch geometry of the F(4x4) kernel: at 448 x 512 the counts are 7 * 2^k -- a single round of workgroups that
                # leaves CUs to the side branches of the three-stream loop (docs/NOTEBOOK.md 4.5 / 4.6), so the whole-chip fraction above
                # has the occupied-CU fraction beside it
                wgs, ksplit = wino4_launch_shape(dom, B, h, w, conc)
                occ = min(1.0, wgs / 256.0)
                roof['launch'] = {'workgroups': wgs, 'k_split': ksplit, 'cus': 256, 'occupied_cu_frac': round(occ, 4),
                                  'frac_on_occupied_cus': round(ach / PEAK_FP32_MFMA_TFLOPS / occ, 4)}
        else:
            ach = bytes_[dom] / (stage_ms[dom] * 1e-3) / 1e9
            tr, note = pmc_traffic(dom, B)
            roof = {'kernel': dom, 'bound': 'hbm', 'achieved': round(ach, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                    'frac': round(ach / PEAK_HBM_GBS, 4), 'traffic': tr,
                    'bytes_per_launch': bytes_[dom], 'ms_per_launch': stage_ms[dom],
                    'ms_per_launch_between_events': stage_ms_events[dom], 'traffic_source': note}
        # Several loops in flight: a lane's kernel is launched in the shape of a `conc`-times larger batch (fewer, longer workgroups:
        # convc2 84 instead of 168) and the OTHER lanes' kernels fill the rest of the chip.  Alone in a single stream (the replay
        # above) such a launch leaves most CUs idle, so the whole-chip fraction of one instance says little about the product; the
        # figure that does is the same kernel with the chip filled the way the product fills it: `conc` instances on `conc`
        # streams (HIP events on every launch stream, the slowest stream counts).  `roofline` carries that as achieved / frac and
        # keeps the single instance beside it.
        if conc > 1 and roof.get('bound') == 'mfma' and dom in W44_FIELDS and wl.get(dom) == 4.0:
            cin, cout, field = W44_FIELDS[dom]
            us_c = concurrent_wino4_us(model, _dev, _ffi, field, cin, cout, B, h, w, conc)
            ach_c = conc * flops[dom] / wl[dom] / (us_c * 1e-6) / 1e12
            roof['single_instance'] = {k: roof[k] for k in ('achieved', 'frac', 'ms_per_launch', 'ms_per_launch_between_events',
                                                            'achieved_between_events', 'frac_between_events', 'frac_of_sustained_mfma', 'launch') if k in roof}
            roof.update({'achieved': round(ach_c, 2), 'frac': round(ach_c / PEAK_FP32_MFMA_TFLOPS, 4),
                         'frac_of_sustained_mfma': round(ach_c / mfma_sust, 4), 'instances': conc,
                         'ms_per_launch': round(us_c * 1e-3, 5), 'flops_per_launch': flops[dom] / wl[dom],
                         'measured': f'{conc} concurrent instances of the launch on {conc} streams (= the product\'s loops in flight), HIP events on '
                                     'each launch stream around 30 launches, slowest stream; achieved = instances x executed FLOPs per launch / that time'})
            for k in ('ms_per_launch_between_events', 'achieved_between_events', 'frac_between_events'):
                roof.pop(k, None)
        result['roofline'] = roof
        # the HBM-bound kernels the north star singles out: per-launch algorithmic bytes / HIP-event time, against the
        # 8 TB/s datasheet peak (frac) and against this box's measured copy bandwidth (frac_of_measured_copy)
        for name in ('corr_lookup', 'upsample_convex'):
            gbs = bytes_[name] / (stage_ms[name] * 1e-3) / 1e9
            tr, note = pmc_traffic(name, B)
            result['roofline_' + name] = {
                'kernel': name, 'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                'frac': round(gbs / PEAK_HBM_GBS, 4), 'frac_of_measured_copy': round(gbs / copy_gbs, 4),
                'traffic': tr, 'bytes_per_launch': bytes_[name], 'ms_per_launch': stage_ms[name],
                'ms_per_launch_between_events': stage_ms_events[name], 'traffic_source': note}
        # The product loop runs the lookup INSIDE convc1 (raft_lookup_convc1_f32): what the lookup costs there is the fused
        # kernel's time minus the stand-alone convc1's, for the same algorithmic bytes MINUS the 324-channel output that
        # is no longer written or re-read.
        two_us = float(stage_ms['corr_lookup'] + stage_ms['convc1']) * 1e3
        inc_us = max(fused_us - stage_ms['convc1'] * 1e3, 1e-3)
        lk = result['roofline_corr_lookup']
        # NOT a kernel duration: a difference between two different kernels' times (fused kernel - stand-alone convc1).  Kept
        # as what it is -- an estimate of what the lookup adds inside the fused kernel -- and never quoted as a roofline fraction
        lk['in_product_loop_model'] = {
            'kernel': 'lookup fused into convc1 (raft_lookup_convc1_f32)', 'fused_us_per_launch': round(fused_us, 2),
            'two_kernels_us_per_launch': round(two_us, 2), 'standalone_convc1_us': round(stage_ms['convc1'] * 1e3, 2),
            'estimated_incremental_lookup_us': round(inc_us, 2),
            'note': 'fused kernel time minus stand-alone convc1 time: a model, not a measured kernel duration'}
        # ---- the kernels the PRODUCT loop runs in place of the four above, with rooflines of their own.
        # lookup fused into convc1: the 1x1 product (324 real input channels) bounds it -> bound "mfma"; its HBM side beside it
        M_px = B * h * w
        lc_flops, lc_bytes = 2.0 * 324 * 256 * M_px, float(M_px * (4 * 100 * 4 + 8 + 256 * 4))
        tr, note = pmc_traffic('lookup_convc1_fused', B)
        lc_tf = lc_flops / (fused_us * 1e-6) / 1e12
        result['roofline_lookup_convc1_fused'] = {
            'kernel': 'lookup_convc1_kernel (raft_lookup_convc1_f32: pyramid lookup + convc1 1x1 324->256 + relu, what the product loop runs)',
            'bound': 'mfma', 'achieved': round(lc_tf, 2), 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(lc_tf / PEAK_FP32_MFMA_TFLOPS, 4), 'traffic': tr, 'traffic_source': note, 'flops_per_launch': lc_flops,
            'ms_per_launch': round(fused_us * 1e-3, 5),
            'hbm': {'algorithmic_bytes_per_launch': lc_bytes, 'gbs': round(lc_bytes / (fused_us * 1e-6) / 1e9, 1),
                    'frac_of_peak': round(lc_bytes / (fused_us * 1e-6) / 1e9 / PEAK_HBM_GBS, 4),
                    'frac_of_measured_copy': round(lc_bytes / (fused_us * 1e-6) / 1e9 / copy_gbs, 4),
                    'note': 'footprints + coords read, cor1 written; the 324-channel lookup output is neither written nor re-read'}}
        # mask.2 + upsampling (one kernel, the mask never stored): the mask2 stage of the fused replay holds the fused kernel, the
        # upsample stage is an empty bracket
        im, iu = STAGES.index('mask2'), STAGES.index('upsample_convex')
        mu_us = max(float(fused_ms[im] + fused_ms[iu] - 2 * bracket_ms), 1e-3) * 1e3
        mu_bytes = bytes_['upsample_convex'] - 4.0 * B * h * w * 576 + 4.0 * B * h * w * 256     # no mask read; mask.0's output read
        tr, note = pmc_traffic('mask_upsample_fused', B)
        mu_tf = flops['mask2'] / (mu_us * 1e-6) / 1e12
        result['roofline_mask_upsample_fused'] = {
            'kernel': 'mask_upsample_kernel (mask.2 1x1 256->576 + softmax + convex upsampling, what the product loop runs)',
            'bound': 'mfma', 'achieved': round(mu_tf, 2), 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(mu_tf / PEAK_FP32_MFMA_TFLOPS, 4), 'traffic': tr, 'traffic_source': note, 'flops_per_launch': flops['mask2'],
            'ms_per_launch': round(mu_us * 1e-3, 5),
            'two_kernels_us_per_launch': round(float(stage_ms['mask2'] + stage_ms['upsample_convex']) * 1e3, 2),
            'hbm': {'algorithmic_bytes_per_launch': mu_bytes, 'gbs': round(mu_bytes / (mu_us * 1e-6) / 1e9, 1),
                    'mask_bytes_not_moved': 8.0 * B * h * w * 576},
            'note': 'timed as a FULL launch in the single-stream replay; in the three-stream loop every iteration but the last runs it as '
                    '32 long-lived background workgroups on the CUs the chain leaves idle (docs/NOTEBOOK.md 4.6)'}
        # The same stand-alone kernel in the single-stream loop at 8 pairs (BASELINE configs[2] per GPU: a 550 MB volume, larger
        # than the 256 MiB Infinity Cache, which the 275 MB volume of 4 pairs is not -- SURVEY 8d asks for B >= 8) and at 16 pairs
        # (1.1 GB; the launch ramp of a 2.5 us empty grid is amortised over twice the bytes).
        def lookup_at(nb):
            rep = (nb + B - 1) // B
            ia = torch.cat([img1, img1.flip(0)] * rep)[:nb] if B >= 2 else img1[:1].expand(nb, -1, -1, -1).contiguous()
            ib = torch.cat([img2, img2.flip(0)] * rep)[:nb] if B >= 2 else img2[:1].expand(nb, -1, -1, -1).contiguous()
            f1, f2 = model.fnet([2 * (ia / 255.0) - 1.0, 2 * (ib / 255.0) - 1.0])
            cn = model.cnet(2 * (ia / 255.0) - 1.0)
            corr_n = CorrBlock(f1, f2, num_levels=4, radius=4)
            st_n = model._get_state(nb, h, w, device)
            up_n = torch.empty((ITERS, nb, H, W, 2), device=device)
            _ffi.set_option('RAFT_LOOKUP_FUSED', 0)
            _ffi.set_option('RAFT_MASK_FUSED', 0)
            buf = (C.c_float * len(STAGES))()
            acc_n = np.zeros(len(STAGES))
            for _ in range(2):
                model._prepare(cn, st_n)
                _ffi.check(_dev.lib().raft_iterate_basic_timed_f32(
                    C.byref(model.update_block.c), _dev.ptr(corr_n._pyr), corr_n._off, nb, h, w, ITERS, C.byref(st_n.c),
                    _dev.ptr(up_n), _dev.stream_ptr(), buf), 'iterate_basic_timed')
                acc_n += np.array(list(buf))
            _ffi.set_option('RAFT_LOOKUP_FUSED', None)
            _ffi.set_option('RAFT_MASK_FUSED', None)
            ln_ms = max(float(acc_n[0]) / (2 * ITERS) - bracket_ms, 1e-6)
            bn = stage_work(nb, h, w)[1]['corr_lookup']
            trn, _ = pmc_traffic('corr_lookup', nb)
            return {'ms_per_launch': round(ln_ms, 5), 'ms_per_launch_between_events': round(float(acc_n[0]) / (2 * ITERS), 5),
                    'bytes_per_launch': bn, 'achieved': round(bn / (ln_ms * 1e-3) / 1e9, 1), 'unit': 'GB/s',
                    'frac': round(bn / (ln_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                    'frac_of_measured_copy': round(bn / (ln_ms * 1e-3) / 1e9 / copy_gbs, 4), 'traffic': trn}
        if world == 1 and B != 8:
            lk['at_8_pairs'] = lookup_at(8)
        if world == 1 and B != 16:
            lk['at_16_pairs'] = lookup_at(16)
            model._state = None                     # drop the 16-pair state buffers
            torch.cuda.empty_cache()
        # the headline flag is the one at the batches BASELINE's configs use (configs[1]: 4 pairs, configs[2]: 8 per
        # GPU); 16 pairs is reported beside them and earns nothing by itself.
        by_batch = {B: lk['frac_of_measured_copy']}
        for nb in (8, 16):
            if f'at_{nb}_pairs' in lk:
                by_batch[nb] = lk[f'at_{nb}_pairs']['frac_of_measured_copy']
        baseline_batches = [nb for nb in (4, 8) if nb in by_batch]
        lk['target'] = {
            'north_star': '>= 0.60 of the measured HBM copy rate on algorithmic bytes, by kernel duration',
            'frac_of_measured_copy_by_batch': {str(k): v for k, v in sorted(by_batch.items())},
            'target_met_by_batch': {str(k): bool(v >= 0.60) for k, v in sorted(by_batch.items())},
            'baseline_batches': baseline_batches,
            'target_met': bool(baseline_batches) and all(by_batch[nb] >= 0.60 for nb in baseline_batches),
            'floor': 'PMC traffic is 1.32x the algorithmic bytes: a 10x10 footprint touches 6.9 128-byte lines of a 4x8-tiled map (3.1 '
                     'lines of useful floats), and the tile-shape study (profiles/r10b_lookup_layouts_*.txt: 8x4, 2x16, 4x4, 8x8, 2x8, '
                     'row-major; FETCH_SIZE / TCC_EA0_RDREQ per shape) shows the fetch granule is the 128-byte line and no shape pulls '
                     'fewer lines; a 2.5 us empty launch is the rest of the gap at small batches (DESIGN.md section 5); in the product '
                     'loop the kernel runs fused into convc1 and does not write its 18.6 MB (4 pairs) output'}
        # corr_build = pooled-fmap2 pyramid (2 small launches) + ONE fp32-MFMA NT GEMM fmap1 . pyramid^T whose epilogue
        # writes all 4 levels.  Its floor is the GEMM (real FLOPs: every stored correlation value is a C-long dot
        # product), the HBM write of the volume sits below it -- both are reported, bound = "mfma".
        n_vals = sum((h >> l) * (w >> l) for l in range(4)) * h * w * B        # stored correlation values, 4 levels
        build_flops = 2.0 * 256 * n_vals
        build_bytes = B * (2 * h * w * 256 * 4) + 4 * (corr._off[4])           # fmaps read + tiled pyramid written
        bms = pre_ms['corr_build'] * 1e-3
        tr, note = pmc_traffic('corr_build', B)
        result['roofline_corr_build'] = {
            'kernel': 'corr_build (fmap2 pyramid + corr_gemm)', 'bound': 'mfma',
            'achieved': round(build_flops / bms / 1e12, 2), 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(build_flops / bms / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4), 'traffic': tr, 'traffic_source': note,
            'flops_per_launch': build_flops, 'ms_per_launch': round(pre_ms['corr_build'], 4),
            'hbm_gbs': round(build_bytes / bms / 1e9, 1), 'hbm_frac': round(build_bytes / bms / 1e9 / PEAK_HBM_GBS, 4),
            'hbm_frac_of_measured_copy': round(build_bytes / bms / 1e9 / copy_gbs, 4), 'bytes_per_launch': build_bytes}
        # ---- the rocprofv3 side of every roofline object: average kernel durations of the same single-stream launches from the
        # committed trace summary (profiles/kernel_durations.json <- tools/closing_set.sh), flagged stale when it was taken on
        # other kernel sources than this process runs; `hip_events_over_rocprof` is how far the live figure is from it
        durs, dur_ev = evidence_file('kernel_durations.json')
        result['evidence'] = {'kernel_durations': dur_ev, 'pmc_traffic': evidence_file('pmc_traffic.json')[1],
                              'pmc_traffic_at_this_batch': evidence_file(f'pmc_traffic_b{B}.json')[1]}

        def attach(obj, key, work, peak, scale, batch=B):
            us = rocprof_us(durs, batch, key)
            if us is None:
                obj['rocprof'] = None
                return
            a = work / (us * 1e-6) / scale
            obj['rocprof'] = {'us_per_launch': us, 'achieved': round(a, 2), 'frac': round(a / peak, 4), 'stale': dur_ev['stale'],
                              'hip_events_over_rocprof': round(obj['ms_per_launch'] * 1e3 / us, 3), 'batch': batch}
        r = result['roofline']
        if 'instances' in r:
            # measured with `instances` concurrent launches: the rocprofv3 side is the kernel trace of the same microbenchmark
            # (tools/concurrent_kernel.py under rocprofv3 -> profiles/kernel_concurrent.json); the single instance against the single-stream trace
            cdur, cev = evidence_file('kernel_concurrent.json')
            ent = (cdur or {}).get(r['kernel'])
            if ent and ent.get('batch') == B and ent.get('instances') == r['instances']:
                a = r['instances'] * r['flops_per_launch'] / (ent['avg_us'] * 1e-6) / 1e12
                r['rocprof'] = {'us_per_launch': ent['avg_us'], 'achieved': round(a, 2), 'frac': round(a / r['peak'], 4), 'stale': cev['stale'],
                                'hip_events_over_rocprof': round(r['ms_per_launch'] * 1e3 / ent['avg_us'], 3), 'batch': B, 'instances': ent['instances'],
                                'source': 'profiles/kernel_concurrent.json: average kernel duration while `instances` launches share the chip'}
            else:
                r['rocprof'] = None
            result['evidence']['kernel_concurrent'] = cev
            attach(r['single_instance'], r['kernel'], r['flops_per_launch'], r['peak'], 1e12)
        else:
            attach(r, r['kernel'], r.get('flops_per_launch', r.get('bytes_per_launch')), r['peak'], 1e12 if r['bound'] == 'mfma' else 1e9)
        for name in ('corr_lookup', 'upsample_convex'):
            attach(result['roofline_' + name], name, bytes_[name], PEAK_HBM_GBS, 1e9)
            if result['roofline_' + name]['rocprof']:
                result['roofline_' + name]['rocprof']['frac_of_measured_copy'] = round(result['roofline_' + name]['rocprof']['achieved'] / copy_gbs, 4)
        for nb in (8, 16):
            key = f'at_{nb}_pairs'
            if key in lk:
                attach(lk[key], 'corr_lookup', lk[key]['bytes_per_launch'], PEAK_HBM_GBS, 1e9, batch=nb)
                if lk[key]['rocprof']:
                    lk[key]['rocprof']['frac_of_measured_copy'] = round(lk[key]['rocprof']['achieved'] / copy_gbs, 4)
        attach(result['roofline_lookup_convc1_fused'], 'lookup_convc1_fused', lc_flops, PEAK_FP32_MFMA_TFLOPS, 1e12)
        attach(result['roofline_mask_upsample_fused'], 'mask_upsample_fused', flops['mask2'], PEAK_FP32_MFMA_TFLOPS, 1e12)
        attach(result['roofline_corr_build'], 'corr_build', build_flops, PEAK_FP32_MFMA_TFLOPS, 1e12)
        mfma_ms = sum(stage_ms[k] for k in flops)
        result['update_block_tflops'] = round(sum(flops.values()) / (mfma_ms * 1e-3) / 1e12, 2)
        result['update_block_executed_tflops'] = round(
            sum(v / wl.get(k, 1.0) for k, v in flops.items())
            / (mfma_ms * 1e-3) / 1e12, 2)
        result['winograd_layers'] = {k: WINOGRAD_ALGORITHMS[wl[k]] for k in sorted(wl)}
        result['stage_ms'] = stage_ms
        result['stage_ms_between_events'] = stage_ms_events
        result['event_bracket_us'] = round(bracket_ms * 1e3, 3)
        result['single_stream_iteration_ms'] = {'sum_of_bracketed_stages': round(float(per_launch_ms.sum()), 5),
                                                'same_launches_without_events': round(plain_iter_ms, 5)}
        result['pre_loop_ms'] = {k: round(v, 4) for k, v in pre_ms.items()}
        result['roofline_timing'] = ('HIP events on the launch stream, instrumented replay of the timed steps; per-stage '
                                     'times = interval between events minus event_bracket_us (calibrated in the same run against '
                                     'the un-bracketed single-stream loop: the net stages add up to its measured iteration time)')

        _dev.lib().raft_set_thread_concurrency(1)
        # ---------------- PCIe-inclusive rate (host-resident fp32 inputs), informational only: never `value`
        if world == 1:
            from tf_raft_amd.prefetch import prefetch_to_device
            h1, h2 = img1.cpu().numpy(), img2.cpu().numpy()
            n = 30                                  # batches per feed: the first upload of a feed is not hidden

            def fed(feed):
                for a, b in feed(2):
                    step(a, b)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for a, b in feed(n):
                    step(a, b)
                torch.cuda.synchronize()
                return round(B * n / (time.perf_counter() - t0), 3)
            # pageable arrays handed straight to the model (synchronous upload in front of every step) ...
            result['value_incl_h2d_blocking'] = fed(lambda k: ((h1, h2) for _ in range(k)))
            # ... and through the prefetch stage: pinned staging, upload of batch i+1 under the compute of batch i
            result['value_incl_h2d'] = fed(lambda k: prefetch_to_device(((h1, h2) for _ in range(k)), buffer_size=1))

        # ---------------- the EPE half of the metric + the CPU baseline: the oracle (reference restatement) on this box's host
        # cores, ONE (1,448,512,3) pair (element 0 of the timed batch), one timed forward per weight regime (same arithmetic)
        if (world == 1 and not args.no_cpu_baseline) or (world > 1 and regime_pred0):
            import oracle
            c1, c2 = img1[:1].cpu().numpy(), img2[:1].cpu().numpy()
            ncpu = os.cpu_count() or 1
            cands = sorted({n for n in (8, 16, 32, 64, 128, torch.get_num_threads()) if n <= ncpu})
            threads = best_oracle_threads(lambda: oracle.RAFT(wts, iters_pred=2)([c1, c2]), cands)
            times, parity = [], {}
            regs = list(regime_pred0) or ['default']
            for i in range(max(len(regs), args.cpu_runs)):
                reg = regs[i] if i < len(regs) else 'default'
                o = oracle.RAFT(regime_weights.get(reg, wts), iters_pred=ITERS)
                t0 = time.perf_counter()
                want = o([c1, c2])
                times.append(time.perf_counter() - t0)
                if i < len(regs) and reg in regime_pred0:
                    parity[reg] = parity_stats(regime_pred0[reg], want)
            if world == 1:
                result['cpu_baseline'] = {
                    'value': round(1.0 / float(np.median(times)), 4), 'unit': 'image-pairs/s',
                    'cores': threads, 'host_cpus': ncpu, 'kind': 'port', 'runs_s': [round(t, 2) for t in times],
                    'sample': f'{len(times)} x (1,{H},{W},3) pair, iters_pred={ITERS}, torch-CPU fp32 restatement of the tf.keras path '
                              f'(oracle/), one forward per weight regime {regs} (same arithmetic), median; {threads} threads = the fastest '
                              f'of {cands} on a 2-iteration probe; TensorFlow itself is not installable here'}
            if parity:
                # the top-level EPE fields describe the TIMED configuration (`value` is measured on the Keras-default
                # weights); the regimes where the 1e-3 bound is provable are reported under their own name beside them.
                prov = [r for r in ('conditioned', 'jump0') if r in parity]
                dflt = parity.get('default')
                result['final_iter_epe'] = dflt['final_iter_epe'] if dflt else None
                result['final_iter_epe_regime'] = 'default (the weights `value` is timed on)' if dflt else None
                result['final_iter_epe_within_tolerance'] = bool(dflt and dflt['within_tol_on_every_iteration'])
                if dflt:
                    result['final_iter_epe_default_horizon'] = {
                        'iterations_within_tol': dflt['iterations_within_tol'], 'of': ITERS,
                        'final_frac_pixels_within_tol': dflt.get('final_frac_pixels_within_tol', 1.0)}
                result['final_iter_epe_conditioned'] = max(parity[r]['final_iter_epe'] for r in prov) if prov else None
                result['final_iter_epe_conditioned_regimes'] = prov
                result['final_iter_epe_conditioned_within_tolerance'] = bool(prov) and all(parity[r]['within_tol_on_every_iteration'] for r in prov)
                result['final_iter_epe_by_regime'] = {r: parity[r]['final_iter_epe'] for r in parity}
                result['parity'] = {
                    'tolerance': EPE_TOL, 'compared': 'flow_predictions[0..23] of element 0 of the timed batch (HIP, computed inside the '
                    f'batch of {B}) against the CPU oracle run on that pair alone; max over pixels of the 2-norm of the difference',
                    'reference': 'oracle/ = CPU restatement of the reference, asserted bit-equal to the reference\'s unmodified source run '
                                 'on a stand-in tensorflow (tests/test_reference_under_stub.py); TensorFlow 2.3 itself is not installable '
                                 'here, so the semantics of the TF primitives remain recalled -- DESIGN.md section 2',
                    'regimes': parity,
                    'note': 'default = the weights `value` is timed on: an untrained RAFT is ill conditioned there (any two fp32 evaluations '
                            'part ways at the first flipped tap), so it is reported by horizon and locality; conditioned / jump0 are the '
                            'regimes where the 1e-3 bound is provable: final_iter_epe_conditioned is their worst.  final_iter_epe (top level) is the '
                            'default regime, i.e. the timed configuration, and is NOT within tolerance there'}
        if regime_rate:
            result['pairs_per_s_by_regime'] = regime_rate
            lo, hi = min(regime_rate.values()), max(regime_rate.values())
            result['regime_spread_frac'] = round((hi - lo) / hi, 4)
            result['pairs_per_s_by_regime_rounds'] = {r: [round(v, 1) for v in vs] for r, vs in regime_rounds.items()}
            result['regime_timing'] = (f'{len(next(iter(regime_rounds.values())))} interleaved rounds of {args.steps} steps per regime in this '
                                       'process (same inputs, same launches; only the weights differ), median per regime; `value` is the '
                                       'separately timed headline on the default weights')
        front = ['metric', 'value', 'unit', 'final_iter_epe', 'final_iter_epe_regime', 'final_iter_epe_within_tolerance',
                 'final_iter_epe_default_horizon', 'final_iter_epe_conditioned', 'final_iter_epe_conditioned_regimes',
                 'final_iter_epe_conditioned_within_tolerance', 'final_iter_epe_by_regime', 'pairs_per_s_by_regime', 'regime_spread_frac']
        result = {**{k: result[k] for k in front if k in result}, **{k: v for k, v in result.items() if k not in front}}
        print(json.dumps(result), flush=True)
        try:
            os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
            with open(os.path.join(ROOT, 'gpurun_out', f'bench_n{world}.json'), 'w') as f:
                json.dump(result, f, indent=1)
        except OSError:
            pass
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
