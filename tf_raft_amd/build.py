"""Builds ``libraft_hip.so`` (gfx950) in-tree with hipcc.

``python -m tf_raft_amd.build`` or ``tf_raft_amd.build.build_library()``.
hipcc cross-compiles without a GPU, so this also runs in the CPU-only container; the
resulting ``tf_raft_amd/lib/libraft_hip.so`` travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIBPATH = os.path.join(LIBDIR, 'libraft_hip.so')
SOURCES = ['corr.hip', 'upsample.hip', 'conv.hip', 'conv_halo_11.hip', 'conv_halo_33.hip', 'conv_halo_15.hip',
           'conv_halo_51.hip', 'encoder.hip', 'ondemand.hip', 'host_util.hip', 'conv_wino.hip', 'conv_wino1d.hip', 'conv_wino4.hip', 'mask_upsample.hip', 'metrics.hip', 'backward.hip']
HEADERS = ['common.h', 'conv_mfma.h', 'conv_halo.h', 'conv_wino.h', 'conv_wino1d.h', 'conv_wino4.h', 'lookup_common.h', os.path.join('..', '..', 'include', 'raft_hip.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=on',
         '-Wall', '-Wno-unused-function']
# diagnostic builds (e.g. RAFT_BUILD_DEFINES=-DRAFT_FUSED_PROBE for tools/fused_probe.py): part of the digest, so the
# next ordinary load rebuilds the product library
FLAGS += os.environ.get('RAFT_BUILD_DEFINES', '').split()


def _hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found: cannot build libraft_hip.so')
    return exe


def _digest(paths):
    h = hashlib.sha256()
    h.update(' '.join(FLAGS).encode())
    for p in paths:
        with open(p, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def _inputs():
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    return srcs, hdrs


def source_digest() -> str:
    """Digest of the sources + flags as they are on disk now."""
    srcs, hdrs = _inputs()
    return _digest(srcs + hdrs)


def built_digest():
    """Digest the existing libraft_hip.so was built from (None: no library or no stamp)."""
    stamp = os.path.join(LIBDIR, 'build.sha256')
    if not (os.path.exists(LIBPATH) and os.path.exists(stamp)):
        return None
    with open(stamp) as f:
        return f.read().strip()


def build_library(force: bool = False, verbose: bool = True) -> str:
    """Build (or reuse) the library.  Safe against concurrent callers -- e.g. the 8 ranks of a multi-GPU launch on a box
    without a prebuilt .so: an inter-process file lock serialises them, the first builds, the others reuse; the shared
    object is linked under a temporary name and renamed into place."""
    import fcntl
    os.makedirs(LIBDIR, exist_ok=True)
    with open(os.path.join(LIBDIR, '.build.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


# True after a call of build_library() in this process that compiled (False: the existing library was up to date).
# bench.py's multi-GPU preflight reports it per rank: on a box that received the prebuilt .so no rank should compile.
LAST_BUILD_COMPILED = False


def _build_locked(force: bool, verbose: bool) -> str:
    global LAST_BUILD_COMPILED
    srcs, hdrs = _inputs()
    stamp = os.path.join(LIBDIR, 'build.sha256')
    digest = _digest(srcs + hdrs)
    if not force and os.path.exists(LIBPATH) and os.path.exists(stamp):
        with open(stamp) as f:
            if f.read().strip() == digest:
                return LIBPATH
    hipcc = _hipcc()
    objs = []

    def compile_one(src):
        obj = os.path.join(LIBDIR, os.path.basename(src) + '.o')
        cmd = [hipcc, *FLAGS, '-x', 'hip', '-c', src, '-o', obj]
        if verbose:
            print('[build]', ' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, srcs))
    tmp = LIBPATH + f'.tmp{os.getpid()}'
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', tmp]
    if verbose:
        print('[build]', ' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(tmp, LIBPATH)
    with open(stamp, 'w') as f:
        f.write(digest)
    LAST_BUILD_COMPILED = True
    return LIBPATH


def build_sanitizer_library(kind: str = 'address', verbose: bool = False) -> str:
    """HOST-ONLY build of the library under a sanitizer (``kind`` = 'address' or 'thread'):
    ``hipcc --offload-host-only -fsanitize=<kind>`` of the same sources -> ``lib/libraft_hip_host_<kind>.so``.  It holds no
    device code (GPU AddressSanitizer needs xnack+ code objects, which this pool refuses) and is never loaded by the product:
    tests/test_abi_sanitizers.py links ``tests/native/abi_host_check.cpp`` against it to run everything an entry point does
    before it launches -- argument validation, geometry arithmetic, the option table -- from several host threads."""
    if kind not in ('address', 'thread'):
        raise ValueError(f"kind must be 'address' or 'thread', got {kind!r}")
    srcs, hdrs = _inputs()
    out = os.path.join(LIBDIR, f'libraft_hip_host_{kind}.so')
    stamp = out + '.sha256'
    digest = _digest(srcs + hdrs) + ':' + kind
    if os.path.exists(out) and os.path.exists(stamp) and open(stamp).read().strip() == digest:
        return out
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    flags = ['--offload-host-only', f'-fsanitize={kind}', '-O1', '-g', '-fno-omit-frame-pointer', '-std=c++17', '-fPIC',
             '-ffp-contract=on', '-Wno-unused-function', '-Wno-unused-value']

    def compile_one(src):
        obj = os.path.join(LIBDIR, os.path.basename(src) + f'.host_{kind}.o')
        cmd = [hipcc, *flags, '-x', 'hip', '-c', src, '-o', obj]
        if verbose:
            print('[build]', ' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, srcs))
    # a host-only object still refers to the device image its kernels would live in (__hip_fatbin_<hash>): give every such
    # symbol an empty definition -- nothing here ever launches
    syms = set()
    for o in objs:
        for line in subprocess.run(['nm', '-u', o], check=True, capture_output=True, text=True).stdout.splitlines():
            name = line.split()[-1] if line.split() else ''
            if name.startswith('__hip_fatbin_'):
                syms.add(name)
    stub_c = os.path.join(LIBDIR, f'fatbin_stubs_{kind}.c')
    with open(stub_c, 'w') as f:
        f.write('/* generated by tf_raft_amd/build.py build_sanitizer_library */\n')
        for name in sorted(syms):
            f.write(f'const char {name}[64] __attribute__((aligned(4096))) = {{0}};\n')
    stub_o = stub_c[:-2] + '.o'
    subprocess.run(['gcc', '-fPIC', '-c', stub_c, '-o', stub_o], check=True)
    subprocess.run([hipcc, '--offload-host-only', f'-fsanitize={kind}', '-shared-libsan', '-shared', '-fPIC', *objs, stub_o, '-o', out], check=True)
    for o in objs + [stub_c, stub_o]:
        os.remove(o)
    with open(stamp, 'w') as f:
        f.write(digest)
    return out


def build_abi_host_check(kind: str = 'address', verbose: bool = False) -> str:
    """``tests/native/abi_host_check.cpp`` linked against the host-only sanitizer build; returns the executable."""
    lib = build_sanitizer_library(kind, verbose)
    root = os.path.normpath(os.path.join(HERE, '..'))
    src = os.path.join(root, 'tests', 'native', 'abi_host_check.cpp')
    exe = os.path.join(LIBDIR, f'abi_host_check_{kind}')
    clang = os.path.join(os.path.dirname(os.path.realpath(_hipcc())), '..', 'lib', 'llvm', 'bin', 'clang++')
    if not os.path.exists(clang):
        clang = '/opt/rocm/lib/llvm/bin/clang++'
    cmd = [clang, f'-fsanitize={kind}', '-shared-libsan', '-O1', '-g', '-std=c++17', '-pthread', '-I', os.path.join(root, 'include'), src,
           lib, f'-Wl,-rpath,{LIBDIR}', '-o', exe]
    if verbose:
        print('[build]', ' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return exe


if __name__ == '__main__':
    if '--sanitize' in sys.argv:
        for k in ('address', 'thread'):
            print(build_abi_host_check(k, verbose=True))
        sys.exit(0)
    build_library(force='--force' in sys.argv)
    print(LIBPATH)
