"""Register / scratch / LDS account of every gfx950 kernel in the built objects (csrc/build/*.o).

    python scripts/kernel_resources.py [--filter SUBSTR] [--spills-only] [--json]

The device code object of an object file sits in its `.hip_fatbin` section as an offload bundle:
llvm-objcopy dumps the section, clang-offload-bundler unbundles the gfx950 ELF, llvm-readelf prints the
AMDGPU metadata note (per kernel: .vgpr_count, .sgpr_count, .vgpr_spill_count, .sgpr_spill_count,
.private_segment_fixed_size = scratch bytes per lane, .group_segment_fixed_size = static LDS).
tests/test_code_objects.py uses `kernels()` to keep the hot instantiations free of scratch.
"""
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, 'backpacks-flash-attn_amd', 'csrc', 'build')
LLVM = '/opt/rocm/lib/llvm/bin'
TARGET = 'hipv4-amdgcn-amd-amdhsa--gfx950'
FIELDS = ('vgpr_count', 'agpr_count', 'sgpr_count', 'vgpr_spill_count', 'sgpr_spill_count',
          'private_segment_fixed_size', 'group_segment_fixed_size', 'max_flat_workgroup_size')


def tools_available():
    return all(os.path.exists(os.path.join(LLVM, t)) for t in ('llvm-objcopy', 'clang-offload-bundler', 'llvm-readelf'))


def _run(*cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('%s failed:\n%s' % (cmd[0], r.stderr[-2000:]))
    return r.stdout


def code_object_notes(obj):
    """Metadata note text of the gfx950 code object embedded in `obj`."""
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, 'fat.bin'), os.path.join(tmp, 'dev.co')
        if '.hip_fatbin' not in _run(os.path.join(LLVM, 'llvm-readelf'), '-S', obj):
            return ''   # host-only object (bp_api.o)
        _run(os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, obj)
        _run(os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', '--targets=' + TARGET,
             '--input=' + fat, '--output=' + co)
        return _run(os.path.join(LLVM, 'llvm-readelf'), '--notes', co)


def waves_per_simd(vgprs, agprs=0):
    """Register-limited waves per SIMD (512 registers per lane, granule 8; MI355X_MICROARCH 'Register files')."""
    alloc = -(-max(vgprs + agprs, 1) // 8) * 8
    return min(8, 512 // alloc)


def kernels(objects=None):
    """[{name (demangled), mangled, object, <FIELDS>...}] for every kernel of the given objects (default: all built)."""
    objects = objects or sorted(glob.glob(os.path.join(BUILD, '*.o')))
    out = []
    for obj in objects:
        cur = None
        for line in code_object_notes(obj).splitlines():
            # a kernel entry of `amdhsa.kernels` opens with "  - .agpr_count:" (keys are sorted); its own keys sit at
            # four spaces, the keys of its .args entries deeper
            m = re.match(r'^  (- | {2})\.(\w+):\s+(.*)$', line)
            if not m:
                continue
            key, val = m.group(2), m.group(3).strip().strip("'")
            if m.group(1) == '- ':
                cur = {'object': os.path.basename(obj)}
                out.append(cur)
            if cur is None:
                continue
            if key in FIELDS:
                cur[key] = int(val)
            elif key == 'symbol':
                cur['mangled'] = val[:-3] if val.endswith('.kd') else val
    names = [k.get('mangled', '?') for k in out]
    if names:
        import shutil
        filt = shutil.which('c++filt')
        dem = _run(filt, *names).splitlines() if filt else names
        for k, d in zip(out, dem):
            d = re.sub(r'^void ', '', d)
            k['name'] = re.sub(r'\(.*\)$', '', d).replace('bp::', '')
    for k in out:
        # .vgpr_count is the unified total on gfx90a+ (arch VGPRs + AGPRs after alignment); .agpr_count the AGPR part
        k['waves_per_simd'] = waves_per_simd(k.get('vgpr_count', 0))
    return out


def main(argv):
    flt = argv[argv.index('--filter') + 1] if '--filter' in argv else ''
    ks = [k for k in kernels() if flt in k.get('name', '')]
    if '--spills-only' in argv:
        ks = [k for k in ks if k.get('vgpr_spill_count', 0) or k.get('private_segment_fixed_size', 0)]
    if '--json' in argv:
        print(json.dumps(ks, indent=1))
        return
    print('%-78s %5s %5s %6s %6s %8s %7s %5s' % ('kernel', 'vgpr', 'sgpr', 'vspill', 'sspill', 'scratch', 'lds', 'w/SIMD'))
    for k in ks:
        print('%-78s %5d %5d %6d %6d %8d %7d %5d' % (k.get('name', '?')[:78], k.get('vgpr_count', 0), k.get('sgpr_count', 0),
                                                     k.get('vgpr_spill_count', 0), k.get('sgpr_spill_count', 0),
                                                     k.get('private_segment_fixed_size', 0), k.get('group_segment_fixed_size', 0),
                                                     k['waves_per_simd']))


if __name__ == '__main__':
    main(sys.argv[1:])
