"""bench.py's roofline denominators against the hot-path contract; host-only, no GPU.

The per-unit figures (one image pair at 448 x 512, fp32) are the contract's: corr_lookup 10.4 MB / iteration / pair,
upsample_convex 10.1 MB / iteration / pair, update block 3,118,336 MAC / pixel = 22.35 GFLOP / iteration / pair.  The
bench multiplies them by the pairs per launch; these tests keep the two in step and pin the bookkeeping around them
(Winograd multiply ratios, the scaling of the committed PMC pass to the run's batch)."""
import json
import os

import pytest

import bench


def test_algorithmic_work_per_launch_is_the_contract_times_the_batch():
    h, w = bench.H // 8, bench.W // 8
    assert (h, w) == (56, 64)
    flops1, bytes1 = bench.stage_work(1, h, w)
    flops4, bytes4 = bench.stage_work(4, h, w)
    n = h * w
    # lookup = 4 levels x N x (2r+2)^2 x 4 B read + coords + N x 324 x 4 B written = 10.4 MB / iter / pair
    assert bytes1['corr_lookup'] == n * (4 * 100 * 4 + 8 + 324 * 4)
    assert bytes1['corr_lookup'] / 1e6 == pytest.approx(10.4, abs=0.05)
    # upsample: mask 576 + flow 2 floats read, 8 x 8 x 2 floats written per coarse pixel = 10.1 MB / iter / pair
    assert bytes1['upsample_convex'] / 1e6 == pytest.approx(10.1, abs=0.05)
    # update block: 3,118,336 MAC / px, of which the loop-invariant GRU context rows (4 x 5 x 128 x {256, 128, 256, 128}
    # = 491,520 MAC / px) are evaluated once per forward, not per iteration
    macs_per_px = sum(flops1.values()) / (2.0 * n)
    assert macs_per_px + 5 * 128 * (256 + 128 + 256 + 128) == 3118336
    assert 2 * 3118336 * n / 1e9 == pytest.approx(22.35, abs=0.01)
    for k in flops1:                                            # per launch = per pair x pairs per launch
        assert flops4[k] == 4 * flops1[k]
    for k in bytes1:
        assert bytes4[k] == 4 * bytes1[k]
    assert set(flops1) | set(bytes1) == set(bench.STAGES)       # every stage of the loop is priced exactly once
    assert not set(flops1) & set(bytes1)


def test_winograd_layers_follow_the_library_switches():
    """roofline.achieved counts EXECUTED MFMA FLOPs: direct FLOPs / {2.25, 2.5, 10/6} for the layers that are on a Winograd
    kernel under the current switches, so that frac <= 1 (the headline frac was 1.30)."""
    from tf_raft_amd import _ffi
    saved = {k: _ffi.get_option(k) for k in ('RAFT_CONV_WINO', 'RAFT_CONV_WINO4', 'RAFT_GRU_WINO', 'RAFT_GRU_WINO4')}
    try:
        for k in saved:
            _ffi.set_option(k, None)
        d = bench.winograd_layers()
        assert d == {'convc2': 4.0, 'convf2': 4.0, 'conv': 2.25, 'fh1_mask0': 4.0, 'gru_zr1': 2.5, 'gru_q1': 2.5, 'gru_zr2': 2.5,
                     'gru_q2': 2.5}                               # F(4x4,3x3): the flow / mask head, convc2 and convf2 by default ...
        assert bench.winograd_layers(2) == {**{k: v for k, v in d.items() if k not in ('convc2', 'convf2')}, 'convc2': 2.25}
        assert bench.winograd_layers(1)['fh1_mask0'] == 2.25      # a single pair: F(2x2) everywhere
        assert bench.winograd_layers(8)['conv'] == 4.0            # ... and conv from 8 pairs per launch on
        _ffi.set_option('RAFT_CONV_WINO4', 0)
        assert bench.winograd_layers(8)['conv'] == 2.25 and bench.winograd_layers()['fh1_mask0'] == 2.25
        _ffi.set_option('RAFT_GRU_WINO4', 0)
        assert bench.winograd_layers()['gru_zr1'] == pytest.approx(10.0 / 6.0)
        _ffi.set_option('RAFT_GRU_WINO', 0)
        _ffi.set_option('RAFT_CONV_WINO', 8)
        assert bench.winograd_layers() == {'fh1_mask0': 2.25}
        assert set(bench.WINOGRAD_ALGORITHMS) == {2.25, 4.0, 2.5, 10.0 / 6.0}
    finally:
        for k, v in saved.items():
            _ffi.set_option(k, v if v != '' else None)


def test_pmc_traffic_scales_the_committed_pass_per_pair():
    """`traffic` = HBM bytes per launch from the committed in-loop PMC pass (profiles/pmc_traffic.json), per pair x the
    run's batch; the pass must be the in-loop B >= 8 one the round-1 review asked for and cover every roofline kernel."""
    with open(os.path.join(bench.ROOT, 'profiles', 'pmc_traffic.json')) as f:
        table = json.load(f)
    for kernel in ('fh1_mask0', 'corr_lookup', 'upsample_convex', 'corr_build'):
        entry = table[kernel]
        assert entry['batch'] >= 8
        tr4, note = bench.pmc_traffic(kernel, 4)
        tr8, _ = bench.pmc_traffic(kernel, 8)
        tr2, note2 = bench.pmc_traffic(kernel, 2)            # no pass at 2 pairs: the 8-pair pass scaled per pair
        assert tr8 == pytest.approx(entry['hbm_bytes_per_launch'] * 8 / entry['batch'], abs=1)
        assert tr2 == pytest.approx(tr8 / 4, abs=1) and note2['pass_batch'] == entry['batch'] and note2['scaled_to_batch'] == 2
        # round 6: a pass taken AT the run's batch (profiles/pmc_traffic_b4.json) is preferred over scaling -- unless it is stale
        b4, ev4 = bench.evidence_file('pmc_traffic_b4.json')
        if b4 and not ev4['stale'] and kernel in b4:
            assert note['pass_batch'] == 4 and tr4 == b4[kernel]['hbm_bytes_per_launch']
        else:
            assert tr4 == pytest.approx(tr8 / 2, abs=1) and note['pass_batch'] == entry['batch']
        assert note['scaled_to_batch'] == 4
        if kernel != 'corr_build':
            assert note['in_loop'] is True
        # traffic never below the algorithmic bytes of the kernel (a pass that under-counts would flatter the kernel)
        assert note['hbm_bytes_per_pair'] >= 0.99 * note['algorithmic_bytes_per_pair']
    assert bench.pmc_traffic('no_such_kernel', 4) == (None, None)
    assert bench.PEAK_FP32_MFMA_TFLOPS == 157.3 and bench.PEAK_HBM_GBS == 8000.0


def test_parity_stats_reports_final_epe_horizon_and_locality():
    """The EPE half of the metric (BASELINE.json: "final-iter EPE vs TF ref"): max over pixels of the 2-norm of the difference,
    per iteration; horizon = first iteration beyond 1e-3; a departure is described by how local it is."""
    import numpy as np
    rng = np.random.default_rng(0)
    want = [rng.normal(size=(1, 8, 8, 2)).astype(np.float32) for _ in range(6)]
    got = [w.copy() for w in want]
    for i in range(6):
        got[i][0, 0, 0, 0] += np.float32(1e-5 * (i + 1))                      # slow drift on one pixel: stays inside
    st = bench.parity_stats(got, want)
    assert st['within_tol_on_every_iteration'] and st['iterations_within_tol'] == 6 and 'first_departure' not in st
    assert st['final_iter_epe'] == pytest.approx(6e-5, rel=0.05) and len(st['per_iteration_epe']) == 6
    got[3][0, 2, 2, :] += np.float32(0.3)                                      # a flipped tap at iteration 3: one pixel, 0.42 px
    got[4][0, 2:4, 2:4, :] += np.float32(0.3)
    got[5][0, 2:4, 2:4, :] += np.float32(0.3)
    st = bench.parity_stats(got, want)
    assert not st['within_tol_on_every_iteration'] and st['iterations_within_tol'] == 3
    assert st['first_departure']['iteration'] == 3 and st['first_departure']['frac_pixels_within_tol'] == pytest.approx(63 / 64, abs=1e-4)
    assert st['final_frac_pixels_within_tol'] == pytest.approx(60 / 64, abs=1e-4)
    assert st['final_iter_epe'] == pytest.approx(0.3 * 2 ** 0.5, rel=1e-3)


def test_evidence_files_are_flagged_stale_when_measured_on_other_sources(tmp_path, monkeypatch):
    """profiles/pmc_traffic.json and kernel_durations.json record the digest of the HIP sources they were measured on; the bench
    line must say so when that is not the library it runs (traffic figures from a pass that predated three kernel changes)."""
    from tf_raft_amd import build
    prof = tmp_path / 'profiles'
    prof.mkdir()
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    entry = {'batch': 8, 'in_loop': True, 'hbm_bytes_per_launch': 800, 'algorithmic_bytes_per_pair': 90, 'source': 'x.csv'}
    (prof / 'pmc_traffic.json').write_text(json.dumps({'_meta': {'source_digest': build.source_digest(), 'git_head': 'abc'}, 'convc2': entry}))
    tr, note = bench.pmc_traffic('convc2', 4)
    assert tr == 400 and note['stale'] is False and note['measured_at_commit'] == 'abc' and note['hbm_bytes_per_pair'] == 100
    (prof / 'pmc_traffic.json').write_text(json.dumps({'_meta': {'source_digest': 'digest-of-older-kernels'}, 'convc2': entry}))
    assert bench.pmc_traffic('convc2', 4)[1]['stale'] is True
    (prof / 'pmc_traffic.json').write_text(json.dumps({'convc2': entry}))               # a file from before round 4: no digest
    assert bench.pmc_traffic('convc2', 4)[1]['stale'] is True
    assert bench.pmc_traffic('no_such_stage', 4) == (None, None)
    d, ev = bench.evidence_file('kernel_durations.json')                                 # absent
    assert d is None and ev['present'] is False and ev['stale'] is True
    (prof / 'kernel_durations.json').write_text(json.dumps({'_meta': {'source_digest': build.source_digest()}, 'b4': {'convc2': {'avg_us': 61.4}}}))
    d, ev = bench.evidence_file('kernel_durations.json')
    assert ev['stale'] is False and bench.rocprof_us(d, 4, 'convc2') == 61.4 and bench.rocprof_us(d, 8, 'convc2') is None


def test_f4x4_launch_shape_follows_the_library_rule_and_its_hints():
    """the launch block of the roofline object must come from the rule the library applies (grid size, RAFT_WINO4_KS,
    the per-layer hints RAFT_CONVC2_KS / RAFT_CONVF2_KS), not from the grid size alone."""
    from tf_raft_amd import _ffi
    names = ('RAFT_WINO4_KS', 'RAFT_CONVC2_KS', 'RAFT_CONVF2_KS')
    try:
        for k in names:
            _ffi.set_option(k, None)
        assert bench.wino4_launch_shape('convc2', 4, 56, 64) == (168, True)       # 84 eight-row workgroups < 128: K split
        assert bench.wino4_launch_shape('fh1_mask0', 4, 56, 64) == (224, False)
        assert bench.wino4_launch_shape('convf2', 4, 56, 64) == (56, True)        # 28 eight-row workgroups: K split
        assert bench.wino4_launch_shape('convf2', 8, 56, 64) == (56, False)       # 56 eight-row workgroups: the hint keeps them
        assert bench.wino4_launch_shape('convc2', 8, 56, 64) == (168, False)
        _ffi.set_option('RAFT_CONVC2_KS', 1)
        assert bench.wino4_launch_shape('convc2', 4, 56, 64) == (84, False)
        _ffi.set_option('RAFT_WINO4_KS', 2)                                        # the global switch wins over every hint
        assert bench.wino4_launch_shape('convc2', 4, 56, 64) == (168, True) and bench.wino4_launch_shape('convf2', 8, 56, 64) == (112, True)
    finally:
        for k in names:
            _ffi.set_option(k, None)
