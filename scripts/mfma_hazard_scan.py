"""Scan the gfx950 code objects for MFMA results that are read too early ACROSS A BRANCH.

    python scripts/mfma_hazard_scan.py [--filter SUBSTR] [--verbose]

Why: gfx950 does not interlock "matrix pipe writes a VGPR -> a vector instruction reads it"; the compiler's hazard
recognizer pads the gap with s_nop.  In round 6 a listing of sense_lse_wide_dma_kernel showed the padding missing on a loop
back-edge: the block behind the branch began with `v_max_f32 v59, v3, v3` one slot after the v_mfma that writes v[0:15]
(the s_nop stood three instructions further down).  The stale read only moved a softmax reference maximum, i.e. the result
by one ulp from launch to launch -- found by a repeatability probe, not by a parity test.  The source-level cure is a pin
(`asm volatile("" : "+v"(acc))`) right behind the MFMA run, which makes the compiler pad inside the block; this scanner
keeps every kernel of the library honest about it (tests/test_code_objects.py).

Method: disassemble each object's gfx950 code, and from every v_mfma walk all paths (fall-through and branch targets) until
the required wait states (passes + 3: 11 for the 8-pass 32x32x16, 7 for 4-pass shapes) have elapsed; any instruction on the
way that names a register of the MFMA's destination -- other than as the accumulator operand of another MFMA -- is reported.
An instruction counts one wait state, `s_nop N` N + 1, another MFMA its own passes (the pipe accepts it only then).
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, 'backpacks-flash-attn_amd', 'csrc', 'build')
LLVM = '/opt/rocm/lib/llvm/bin'
TARGET = 'hipv4-amdgcn-amd-amdhsa--gfx950'


def disassemble(obj):
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, 'fat.bin'), os.path.join(tmp, 'dev.co')
        sec = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '-S', obj], capture_output=True, text=True).stdout
        if '.hip_fatbin' not in sec:
            return ''
        subprocess.run([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, obj], check=True)
        subprocess.run([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', '--targets=' + TARGET,
                        '--input=' + fat, '--output=' + co], check=True, capture_output=True)
        return subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', co], capture_output=True, text=True).stdout


# A taken branch is charged its own issue slot plus three for the redirect (the instruction buffer refill takes >= 16
# clocks = 4 slots on GCN/CDNA); without this allowance every early-exit branch behind an MFMA reads as 1-2 slots short.
TAKEN_BRANCH_STATES = 3

REG = re.compile(r'\b([va])(?:(\d+)|\[(\d+):(\d+)\])')


def regs(text):
    out = set()
    for m in REG.finditer(text):
        lo = int(m.group(2) if m.group(2) is not None else m.group(3))
        hi = int(m.group(2) if m.group(2) is not None else m.group(4))
        out.update((m.group(1), r) for r in range(lo, hi + 1))
    return out


def passes(mnemonic):
    m = re.search(r'_(\d+)x(\d+)x(\d+)', mnemonic)
    if not m:
        return 16
    mm, _, kk = int(m.group(1)), int(m.group(2)), int(m.group(3))
    if mm == 32:
        return 8 if kk >= 16 else 16
    if mm == 16:
        return 4 if kk >= 32 else 8
    return 2


def functions(text):
    cur, name = None, None
    for line in text.splitlines():
        m = re.match(r'^[0-9a-f]+ <(\S+)>:$', line)
        if m:
            if cur:
                yield name, cur
            name, cur = m.group(1), []
            continue
        m = re.match(r'^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):', line)
        if m and cur is not None:
            cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
    if cur:
        yield name, cur


TRANS = ('v_exp_', 'v_log_', 'v_rcp_', 'v_rsq_', 'v_sqrt_', 'v_sin_', 'v_cos_')


def scan(name, ins, verbose=False, trans=True):
    """Early reads behind every v_mfma (passes + 3 wait states) and, with `trans`, behind every transcendental: gfx940+
    forwards a transcendental's result to another transcendental only; a plain VALU that reads it needs ONE wait state in
    between (LLVM's hasTransForwardingHazard), and the same only-inside-a-block padding could miss it across a branch."""
    by_addr = {a: i for i, (a, _, _) in enumerate(ins)}
    found = []
    for i, (addr, mn, ops) in enumerate(ins):
        is_mfma = mn.startswith('v_mfma') or mn.startswith('v_smfmac')
        is_trans = trans and mn.startswith(TRANS)
        if not is_mfma and not is_trans:
            continue
        dst = regs(ops.split(',')[0])
        need = passes(mn) + 3 if is_mfma else 1
        seen = set()
        stack = [(i + 1, 0)]
        while stack:
            j, states = stack.pop()
            while j < len(ins) and states < need:
                if (j, states) in seen:
                    break
                seen.add((j, states))
                a2, mn2, ops2 = ins[j]
                parts = [p.strip() for p in ops2.split(',')]
                if mn2.startswith('v_mfma') or mn2.startswith('v_smfmac'):
                    used = regs(','.join(parts[1:3]))      # A and B operands; the accumulator chain is forwarded in hardware
                elif mn2.startswith('s_') or mn2.startswith('ds_') and False:
                    used = set()
                else:
                    used = regs(ops2)
                if is_trans and (mn2.startswith(TRANS) or not mn2.startswith('v_')):
                    used = set()     # forwarded to another transcendental; memory / LDS / scalar readers are interlocked
                if used & dst and not mn2.startswith('s_'):
                    found.append((name, addr, mn, a2, mn2 + ' ' + ops2, states, need))
                    break
                if mn2 == 's_endpgm':
                    break
                step = 1
                if mn2.startswith('v_mfma') or mn2.startswith('v_smfmac'):
                    step = passes(mn2)      # the matrix pipe takes a new instruction only when the previous one's passes are over
                if mn2 == 's_nop':
                    step = int(ops2) + 1
                if mn2 == 's_branch' or mn2.startswith('s_cbranch'):
                    simm = int(parts[0])
                    simm = simm - 65536 if simm >= 32768 else simm
                    tgt = by_addr.get(a2 + 4 + 4 * simm)
                    if tgt is not None:
                        stack.append((tgt, states + 1 + TAKEN_BRANCH_STATES))
                    if mn2 == 's_branch':
                        break
                states += step
                j += 1
    return found


def main():
    flt = sys.argv[sys.argv.index('--filter') + 1] if '--filter' in sys.argv else ''
    objs = [a for a in sys.argv[1:] if a.endswith('.o')] or sorted(glob.glob(os.path.join(BUILD, '*.o')))
    total = 0
    for obj in objs:
        for name, ins in functions(disassemble(obj)):
            if flt and flt not in name:
                continue
            hits = scan(name, ins)
            seen = set()
            for h in hits:
                key = (h[0], h[3])
                if key in seen:
                    continue
                seen.add(key)
                total += 1
                dem = subprocess.run(['c++filt', h[0]], capture_output=True, text=True).stdout.strip()
                print('%s: %s\n    %s at %#x -> read by `%s` at %#x after %d of %d wait states'
                      % (os.path.basename(obj), dem[:110], h[2], h[1], h[4][:70], h[3], h[5], h[6]))
    print('early reads of MFMA / transcendental results: %d' % total)
    return total


if __name__ == '__main__':
    sys.exit(1 if main() else 0)
