"""Register / LDS budgets of the gfx950 kernels, read from the code objects hipcc produced (host-only: no GPU).

Several hot kernels sit at the edge of an occupancy step (the 64-channel Winograd kernels use 255 of 256 VGPRs, the
on-demand lookup needs <= 168 for three workgroups per CU), and a spill or a lost occupancy step shows up as a slower
bench line long before a parity test notices.  This pins: no vector-register (scratch) spills anywhere in the library, no kernel beyond the
160 KB of LDS of a CU, and the budgets the launch rules in the sources rely on."""
import glob
import os
import re
import shutil
import subprocess

import pytest

LLVM = '/opt/rocm/lib/llvm/bin'
TOOLS = [os.path.join(LLVM, t) for t in ('llvm-objcopy', 'clang-offload-bundler', 'llvm-readelf')]


def _kernels(tmp_path):
    from tf_raft_amd import build
    build.build_library(verbose=False)
    lib_dir = os.path.dirname(build.LIB_PATH) if hasattr(build, 'LIB_PATH') else os.path.join(os.path.dirname(build.__file__), 'lib')
    out = {}
    for obj in sorted(glob.glob(os.path.join(lib_dir, '*.hip.o'))):
        fat, co = str(tmp_path / 'fat.bin'), str(tmp_path / 'dev.co')
        for f in (fat, co):
            if os.path.exists(f):
                os.remove(f)
        if subprocess.run([TOOLS[0], '--dump-section', f'.hip_fatbin={fat}', obj], capture_output=True).returncode != 0:
            continue                                            # a translation unit without device code
        subprocess.run([TOOLS[1], '--unbundle', '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', f'--input={fat}',
                        f'--output={co}'], check=True, capture_output=True)
        notes = subprocess.run([TOOLS[2], '--notes', co], check=True, capture_output=True, text=True).stdout
        for blk in notes.split('  - .agpr_count:')[1:]:
            name = re.search(r'\.name:\s+(\S+)', blk).group(1)
            get = lambda k: int(re.search(r'\.%s:\s+(\d+)' % k, blk).group(1))   # noqa: E731
            demangled = subprocess.run([os.path.join(LLVM, 'llvm-cxxfilt'), name], capture_output=True, text=True).stdout.strip() \
                if os.path.exists(os.path.join(LLVM, 'llvm-cxxfilt')) else name
            out[demangled or name] = {'vgpr': get('vgpr_count'), 'vgpr_spill': get('vgpr_spill_count'),
                                      'sgpr_spill': get('sgpr_spill_count'), 'lds': get('group_segment_fixed_size'),
                                      'threads': get('max_flat_workgroup_size'), 'file': os.path.basename(obj)}
    return out


@pytest.mark.skipif(not all(os.path.exists(t) for t in TOOLS) or shutil.which('hipcc') is None and
                    not os.path.exists('/opt/rocm/bin/hipcc'), reason='ROCm LLVM tools not available')
def test_no_kernel_spills_and_the_occupancy_budgets_hold(tmp_path):
    k = _kernels(tmp_path)
    assert len(k) > 150, f'only {len(k)} kernels found in the build products'
    # vector-register spills go to scratch memory (SGPR spills only move into spare VGPR lanes and are not counted here)
    spilled = sorted(n for n, v in k.items() if v['vgpr_spill'])
    assert not spilled, f'kernels with scratch spills: {spilled}'
    assert max(v['lds'] for v in k.values()) <= 160 * 1024

    def pick(pattern):
        hits = {n: v for n, v in k.items() if re.search(pattern, n)}
        assert hits, pattern
        return hits

    # on-demand lookup: three workgroups per CU = 168 VGPRs and a third of the LDS (ondemand.hip)
    for n, v in pick(r'corr_lookup_ondemand_block_kernel').items():
        assert v['vgpr'] <= 168 and v['lds'] * 3 <= 160 * 1024, (n, v)
    # volume lookup: eight workgroups per CU (docs/NOTEBOOK.md 4.3)
    for n, v in pick(r'corr_lookup_strip_kernel').items():
        assert v['vgpr'] <= 64 and v['lds'] * 8 <= 160 * 1024, (n, v)
    # Winograd F(2x2,3x3): two 256-thread workgroups per CU (<= 256 VGPRs, <= 80 KB); the split-K variant one 512-thread one
    for n, v in pick(r'conv_wino_kernel(<|I)').items():
        if v['threads'] == 512:
            assert v['vgpr'] <= 256 and v['lds'] <= 160 * 1024, (n, v)
        else:
            assert v['vgpr'] <= 256 and v['lds'] * 2 <= 160 * 1024, (n, v)
    for n, v in pick(r'conv_wino1d_kernel(<|I)').items():
        assert v['vgpr'] <= 256 and v['lds'] * 2 <= 160 * 1024, (n, v)
    # fused lookup + convc1: 512 threads, two workgroups per CU -> 128 VGPRs
    for n, v in pick(r'lookup_convc1_kernel').items():
        assert v['vgpr'] <= 128 and v['lds'] * 2 <= 160 * 1024, (n, v)
