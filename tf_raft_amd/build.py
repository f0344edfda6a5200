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


if __name__ == '__main__':
    build_library(force='--force' in sys.argv)
    print(LIBPATH)
