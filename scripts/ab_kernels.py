"""Same-box A/B of kernel micro-benchmarks between library builds (BP_HIP_LIB selects the build per subprocess).

    python scripts/ab_kernels.py --libs default,r3k --which mix,bwd,mixbwd --batch 64 [--reps 3] [--out FILE.jsonl]
`default` = bp_hip/libbackpack_hip.so, any other name N = bp_hip/libbackpack_hip_N.so (build_hip.py --variant N);
`N+VAR=VALUE` runs build N with that environment variable set (run-time switches of one build).
The builds run interleaved, `reps` times each, so clock / thermal drift hits them alike; one JSON line per run."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--libs', default='default')
    ap.add_argument('--which', default='flash')
    ap.add_argument('--batch', default='64')
    ap.add_argument('--seq', default='1024')
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--iters', default='20')
    ap.add_argument('--extra', default='')
    ap.add_argument('--out', default='')
    a = ap.parse_args()
    libs = a.libs.split(',')
    rows = []
    for rep in range(a.reps):
        for name in libs:
            env = dict(os.environ)
            name, *settings = name.split('+')      # `lib+VAR=VALUE+...`: the build plus environment switches
            for kv in settings:
                k, v = kv.split('=', 1)
                env[k] = v
            if name != 'default':
                env['BP_HIP_LIB'] = os.path.join(ROOT, 'backpacks-flash-attn_amd', 'bp_hip', 'libbackpack_hip_%s.so' % name)
            else:
                env.pop('BP_HIP_LIB', None)
            for batch in a.batch.split(','):
                cmd = [sys.executable, os.path.join(ROOT, 'scripts', 'bench_kernels.py'), '--which', a.which,
                       '--batch', batch, '--seq', a.seq, '--iters', a.iters] + a.extra.split()
                r = subprocess.run(cmd, env=env, capture_output=True, text=True)
                if r.returncode != 0:
                    print('FAILED', name, r.stderr[-2000:], flush=True)
                    continue
                for line in r.stdout.splitlines():
                    if line.startswith('{'):
                        row = json.loads(line)
                        row.update(lib='+'.join([name] + settings), rep=rep)
                        rows.append(row)
                        print(json.dumps(row), flush=True)
    # summary: best (min) time per (lib, kernel, batch)
    best = {}
    for r in rows:
        key = (r['kernel'], r.get('batch'), r['lib'])
        best[key] = min(best.get(key, 1e9), r['ms'])
    print('--- best of %d' % a.reps)
    for key in sorted(best):
        print('%-40s batch %-5s %-10s %.4f ms' % (key[0], key[1], key[2], best[key]))
    if a.out:
        with open(a.out, 'w') as f:
            for r in rows:
                f.write(json.dumps(r) + '\n')


if __name__ == '__main__':
    main()
