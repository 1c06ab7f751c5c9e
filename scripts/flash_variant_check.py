"""Bit-for-bit comparison of the flash forward (O, LSE) between two builds of the library over a sweep of shapes:
fixed-length and ragged batches, causal or not, head dims 16..128, both dtypes, dropout, launches with more passes than
resident workgroups (the persistent form walks a list).  Each build runs in its own process (BP_HIP_LIB is read once).

    python scripts/flash_variant_check.py --libs default,persist
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'backpacks-flash-attn_amd')]


def cases():
    out = []
    for dt in ('bf16', 'fp16'):
        for d in (16, 40, 48, 64, 80, 128):
            for s, b, h in ((97, 3, 5), (128, 2, 3), (300, 2, 4), (1024, 2, 12), (2048, 1, 3)):
                for causal in (True, False):
                    if dt == 'fp16' and d in (40, 80) and s in (97, 300):
                        continue
                    out.append(dict(kind='fixed', dt=dt, d=d, s=s, b=b, h=h, causal=causal, p=0.0))
    # many passes per resident workgroup
    out.append(dict(kind='fixed', dt='bf16', d=64, s=1024, b=64, h=12, causal=True, p=0.0))
    out.append(dict(kind='fixed', dt='bf16', d=64, s=1024, b=40, h=12, causal=False, p=0.0))
    out.append(dict(kind='fixed', dt='fp16', d=64, s=4096, b=4, h=12, causal=True, p=0.0))
    out.append(dict(kind='fixed', dt='bf16', d=48, s=1024, b=48, h=16, causal=True, p=0.0, lse_only=True))
    out.append(dict(kind='fixed', dt='bf16', d=16, s=1024, b=16, h=64, causal=True, p=0.0, lse_only=True))
    out.append(dict(kind='fixed', dt='bf16', d=80, s=1024, b=40, h=8, causal=True, p=0.0))
    # dropout
    for d in (64, 80, 128):
        out.append(dict(kind='fixed', dt='bf16', d=d, s=1024, b=8, h=12, causal=True, p=0.17))
        out.append(dict(kind='fixed', dt='fp16', d=d, s=300, b=3, h=4, causal=False, p=0.17))
    # ragged batches: unequal lengths, an empty key sequence, cross lengths
    for d in (40, 64, 128):
        for causal in (True, False):
            out.append(dict(kind='varlen', dt='bf16', d=d, h=5, causal=causal, p=0.0, lq=[97, 128, 33, 1, 700, 256],
                            lk=[97, 128, 33, 1, 700, 256]))
            out.append(dict(kind='varlen', dt='fp16', d=d, h=3, causal=causal, p=0.0, lq=[130, 64, 5, 513],
                            lk=[200, 0, 77, 1024]))
    out.append(dict(kind='varlen', dt='bf16', d=64, h=12, causal=True, p=0.0, lq=[1024] * 30 + [1000, 37, 511] * 20,
                    lk=[1024] * 30 + [1000, 37, 511] * 20))
    return out


def run_case(c, torch, bp_hip):
    dt = torch.bfloat16 if c['dt'] == 'bf16' else torch.float16
    g = torch.Generator(device='cuda').manual_seed(hash((c['d'], c.get('s', 0), c['h'])) % (1 << 30))
    h, d = c['h'], c['d']
    if c['kind'] == 'fixed':
        b, s = c['b'], c['s']
        total_q = total_k = b * s
        cu_q = cu_k = None
        mq = mk = s
    else:
        lq, lk = c['lq'], c['lk']
        total_q, total_k = sum(lq), sum(lk)
        cu_q = torch.tensor([0] + list(__import__('itertools').accumulate(lq)), dtype=torch.int32, device='cuda')
        cu_k = torch.tensor([0] + list(__import__('itertools').accumulate(lk)), dtype=torch.int32, device='cuda')
        mq, mk = max(lq), max(lk)
    q = torch.randn(total_q, h, d, device='cuda', dtype=dt, generator=g)
    k = torch.randn(max(total_k, 1), h, d, device='cuda', dtype=dt, generator=g)[:total_k]
    v = torch.randn(max(total_k, 1), h, d, device='cuda', dtype=dt, generator=g)[:total_k]
    rng = torch.tensor([1234, 77], dtype=torch.int64, device='cuda') if c['p'] > 0 else None
    if c.get('lse_only'):
        lse = bp_hip.flash_fwd(q, k, None, None, cu_q, cu_k, mq, mk, d ** -0.5, c['causal'])
        return [lse]
    o = torch.full_like(q, float('nan'))
    lse = bp_hip.flash_fwd(q, k, v, o, cu_q, cu_k, mq, mk, d ** -0.5, c['causal'], dropout_p=c['p'], rng_state=rng)
    return [o, lse]


def dump(path):
    import torch
    import bp_hip
    res = []
    for c in cases():
        res.append([t.cpu() for t in run_case(c, torch, bp_hip)])
        torch.cuda.synchronize()
    torch.save(res, path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--libs', default='default,persist')
    ap.add_argument('--dump', default='')
    a = ap.parse_args()
    if a.dump:
        return dump(a.dump)
    import torch
    paths = []
    for name in a.libs.split(','):
        env = dict(os.environ)
        if name != 'default':
            env['BP_HIP_LIB'] = os.path.join(ROOT, 'backpacks-flash-attn_amd', 'bp_hip', 'libbackpack_hip_%s.so' % name)
        else:
            env.pop('BP_HIP_LIB', None)
        path = '/tmp/flash_variant_%s.pt' % name
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--dump', path], env=env, capture_output=True, text=True)
        if r.returncode != 0:
            print('FAILED', name, r.stderr[-3000:])
            sys.exit(1)
        paths.append(path)
    a_res, b_res = torch.load(paths[0]), torch.load(paths[1])
    bad = 0
    for c, x, y in zip(cases(), a_res, b_res):
        for i, (s, t) in enumerate(zip(x, y)):
            same = torch.equal(s.view(torch.int16 if s.element_size() == 2 else torch.int32),
                               t.view(torch.int16 if t.element_size() == 2 else torch.int32))
            if not same:
                bad += 1
                diff = (s.float() - t.float())
                nan_a, nan_b = int(torch.isnan(s.float()).sum()), int(torch.isnan(t.float()).sum())
                print('DIFF', c, 'tensor', i, 'max|d|', float(diff.nan_to_num(0).abs().max()), 'nan', nan_a, nan_b,
                      'mismatching', int((s.float() != t.float()).sum()))
    print('cases', len(a_res), 'tensors differing', bad)
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
