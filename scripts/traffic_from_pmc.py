"""HBM bytes per launch from a `scripts/gpu_run.sh pmc` summary, written into profiles/traffic.json (what bench.py quotes
as roofline.traffic).  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 reports both in KiB and, on gfx950,
FETCH_SIZE counts a wide coalesced read at half its size (MI355X_MICROARCH.md, section HBM).
    python scripts/traffic_from_pmc.py <summary.txt> <workload> <batch> [source note]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench_name(kernel):
    if kernel.startswith(('bp::sense_mix_dma_kernel', 'bp::sense_mix_kernel', 'bp::sense_mix_wide')):
        return 'sense_mix_kernel'
    if kernel.startswith('bp::sense_lse_wide'):
        return 'flash_fwd_kernel[lse-only,senses]'
    m = re.match(r'bp::flash_fwd(_dma)?_kernel<bp::\w+, \d+, \d+, (true|false)', kernel)
    if m:
        return 'flash_fwd_kernel' if m.group(2) == 'true' else 'flash_fwd_kernel[lse-only,senses]'
    if kernel.startswith('bp::attn_probs_kernel'):
        return 'attn_probs_kernel'
    return None


def main(summary, workload, batch, note=''):
    counters, cur = {}, None
    for line in open(summary):
        if line.startswith('bp::'):
            cur = line.strip()
            counters[cur] = {}
        elif cur and 'avg=' in line:
            parts = line.split()
            counters[cur][parts[0]] = float(parts[-1].split('=')[-1])
    path = os.path.join(ROOT, 'profiles', 'traffic.json')
    table = json.load(open(path)) if os.path.exists(path) else {}
    table.pop('_comment', None)
    table['_source'] = ('rocprofv3 --pmc passes of scripts/bench_kernels.py (scripts/gpu_run.sh pmc; rounds 1-5: scripts/gpu_pmc.sh), per-kernel averages in '
                        'the profiles/*pmc* summary named per entry; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024')
    for kern, c in counters.items():
        name = bench_name(kern)
        if name is None or 'FETCH_SIZE' not in c or 'WRITE_SIZE' not in c:
            continue
        key = f'{workload}/b{batch}/{name}'
        table[key] = int((2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024)
        table.setdefault('_files', {})[key] = note or os.path.basename(summary)
        hit, miss = c.get('TCC_HIT_sum'), c.get('TCC_MISS_sum')
        extra = f', L2 hit {hit / (hit + miss):.0%}' if hit is not None and miss else ''
        print(f'{key}: {table[key] / 1e6:.1f} MB per launch{extra}')
    json.dump(table, open(path, 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main(*sys.argv[1:])
