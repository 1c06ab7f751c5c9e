"""Build libbackpack_hip.so (gfx950) in-tree with hipcc.  No torch, no cmake.

    python backpacks-flash-attn_amd/build_hip.py [--force] [--verbose]

Objects land in csrc/build/, the shared library in bp_hip/libbackpack_hip.so (git-ignored, but it
travels to the GPU box with the repo snapshot).  hipcc cross-compiles without a GPU.
"""
import hashlib
import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT_DIR = os.path.join(CSRC, 'build')
LIB = os.path.join(HERE, 'bp_hip', 'libbackpack_hip.so')
SOURCES = ['flash_fwd.hip', 'flash_fwd_dma.hip', 'flash_bwd.hip', 'sense_mix.hip', 'sense_mix_dma.hip', 'sense_mix_bwd.hip', 'sense_wide.hip', 'sense_wide_dma.hip', 'attn_probs.hip',
           'add_layer_norm.hip', 'xentropy.hip', 'softmax_bwd.hip', 'bias_gelu.hip', 'bp_api.hip']
HEADERS = ['bp_common.h', 'bp_dma.h', 'bp_kernels.h', 'bp_philox.h', os.path.join('..', '..', 'include', 'bp_hip.h')]
# -amdgpu-mfma-vgpr-form: keep MFMA accumulators in VGPRs (gfx950 has one unified file); without it
# hipcc parks them in AGPRs and copies 64+ registers per tile around the softmax (measured +4..10 %).
# -fno-strict-aliasing: the kernels move 16-bit / fp32 data as raw 8/16-byte words
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-fno-strict-aliasing',
         '-mllvm', '-amdgpu-mfma-vgpr-form=1',
         '-Wall', '-Wno-unused-variable', '-Wno-unused-but-set-variable']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError('hipcc not found')


def _stamp():
    h = hashlib.sha256()
    for name in SOURCES + HEADERS:
        with open(os.path.join(CSRC, name), 'rb') as f:
            h.update(f.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=False, extra_flags=(), lib=LIB, tag='', csrc=CSRC):
    """`extra_flags` / `lib` / `tag` build an experimental variant next to the default library
    (A/B measurements: BP_HIP_LIB=<path> selects it at run time); `csrc`: take the sources from another directory
    (e.g. an earlier revision's kernels checked out next to the tree, `--src-dir`)."""
    out_dir = OUT_DIR + tag
    os.makedirs(out_dir, exist_ok=True)
    stamp_file = os.path.join(out_dir, 'stamp.txt')
    stamp = _stamp() + ' '.join(extra_flags)
    if not force and os.path.exists(lib) and os.path.exists(stamp_file):
        if open(stamp_file).read().strip() == stamp:
            return lib
    hipcc = _hipcc()
    t0 = time.time()

    def compile_one(src):
        obj = os.path.join(out_dir, src.replace('.hip', '.o'))
        cmd = [hipcc] + FLAGS + list(extra_flags) + ['-c', os.path.join(csrc, src), '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s' % (src, r.stderr[-8000:]))
        if verbose and r.stderr:
            print(r.stderr[-4000:])
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    r = subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs,
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n' + r.stderr[-8000:])
    with open(stamp_file, 'w') as f:
        f.write(stamp)
    if verbose:
        print('built %s in %.1fs' % (lib, time.time() - t0))
    return lib


if __name__ == '__main__':
    if '--variant' in sys.argv:   # python build_hip.py --variant NAME -- <extra hipcc flags>
        name = sys.argv[sys.argv.index('--variant') + 1]
        extra = sys.argv[sys.argv.index('--') + 1:] if '--' in sys.argv else []
        src = sys.argv[sys.argv.index('--src-dir') + 1] if '--src-dir' in sys.argv else CSRC
        print(build(force=True, verbose=True, extra_flags=extra, tag='_' + name, csrc=os.path.abspath(src),
                    lib=LIB.replace('.so', '_' + name + '.so')))
    else:
        print(build(force='--force' in sys.argv, verbose=True))
