"""the C ABI is "re-entrant and thread-safe" -- checked on the HOST side under AddressSanitizer and
ThreadSanitizer (GPU AddressSanitizer needs xnack+ code objects, which the GPU pool refuses; sanitizers run on CPU builds only).

``tf_raft_amd.build.build_sanitizer_library(kind)`` compiles the SAME sources host-only (``hipcc --offload-host-only
-fsanitize=<kind>``) and ``tests/native/abi_host_check.cpp`` runs everything an entry point does before it launches -- argument
validation of every launching entry point (null / misaligned pointers, bad shapes must come back as error codes), the geometry and
workspace helpers with valid arguments, the option table -- first on one thread, then from four threads at once.  The device side of
re-entrancy (two host threads, two streams, results bit-identical) is tests/test_gpu_kernels.py::test_c_abi_from_two_host_threads.
"""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime_dir():
    hits = glob.glob('/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so')
    return os.path.dirname(hits[0]) if hits else None


@pytest.mark.parametrize('kind', ['address', 'thread'])
def test_host_side_of_the_c_abi_under_sanitizer(kind):
    if shutil.which('hipcc') is None and not os.path.exists('/opt/rocm/bin/hipcc'):
        pytest.skip('no hipcc: the sanitizer build cannot be made here')
    rt = _runtime_dir()
    if rt is None:
        pytest.skip('clang sanitizer runtimes not installed')
    from tf_raft_amd import build
    try:
        exe = build.build_abi_host_check(kind)
    except (subprocess.CalledProcessError, OSError) as exc:
        # an environment without the host-only / sanitizer toolchain pieces (the product build is checked by build() itself)
        pytest.skip(f'sanitizer build not possible here: {exc}')
    env = dict(os.environ)
    env['LD_LIBRARY_PATH'] = rt + os.pathsep + env.get('LD_LIBRARY_PATH', '')
    # the HIP runtime itself is not instrumented and keeps process-lifetime allocations: leak reports would be about it
    env['ASAN_OPTIONS'] = 'detect_leaks=0:abort_on_error=0:halt_on_error=1'
    env['TSAN_OPTIONS'] = 'halt_on_error=1:report_signal_unsafe=0'
    env.pop('LD_PRELOAD', None)
    p = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=600)
    report = (p.stdout + p.stderr)[-4000:]
    assert p.returncode == 0, report
    assert 'abi_host_check: ok' in p.stdout, report
    assert 'Sanitizer' not in p.stderr, report            # "ERROR: AddressSanitizer", "WARNING: ThreadSanitizer"
