"""In-tree build of libgss_hip.so (gfx950) with hipcc.

    python -m pb_chime5_amd.build [--force]

The shared library is written to ``pb_chime5_amd/lib/libgss_hip.so``; it is
git-ignored but travels with the working tree.  hipcc cross-compiles for gfx950
without a GPU being present.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / 'csrc'
LIB_DIR = PKG / 'lib'
LIB_PATH = LIB_DIR / 'libgss_hip.so'
SOURCES = ['gss_api.hip', 'stft.hip', 'wpe.hip', 'cacgmm.hip', 'mvdr.hip']
HEADERS = ['gss_internal.h', 'jacobi.h', 'dense_wave.h', '../../include/gss_hip.h']
# -amdgpu-mfma-vgpr-form: keep MFMA accumulators in VGPRs.  Without it the compiler
# puts loop-carried accumulators in VGPRs but the MFMA destination in AGPRs and copies
# every accumulator there and back (plus a wait for the MFMA result) in each k step of
# the apply / trailing-update loops.  All kernels here need < 256 VGPRs.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
         '-ffp-contract=on', '-fno-fast-math', '-Wall', '-Wno-unused-function',
         '-mllvm', '-amdgpu-mfma-vgpr-form=1']


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (Path(cand).exists() or cand == 'hipcc'):
            return cand
    raise RuntimeError('hipcc not found')


def _stamp():
    h = hashlib.sha256()
    for name in SOURCES + HEADERS:
        h.update((CSRC / name).read_bytes())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    LIB_DIR.mkdir(exist_ok=True)
    stamp_file = LIB_DIR / 'libgss_hip.stamp'
    stamp = _stamp()
    if (not force and LIB_PATH.exists() and stamp_file.exists()
            and stamp_file.read_text() == stamp):
        return LIB_PATH
    hipcc = _hipcc()
    objs = []

    def compile_one(name):
        obj = LIB_DIR / (Path(name).stem + '.o')
        cmd = [hipcc, *FLAGS, '-c', str(CSRC / name), '-o', str(obj)]
        if verbose:
            print(' '.join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f'hipcc failed for {name}:\n{res.stdout}\n{res.stderr}')
        if verbose and res.stderr.strip():
            print(res.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC',
           *map(str, objs), '-o', str(LIB_PATH)]
    if verbose:
        print(' '.join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f'link failed:\n{res.stdout}\n{res.stderr}')
    stamp_file.write_text(stamp)
    return LIB_PATH


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
