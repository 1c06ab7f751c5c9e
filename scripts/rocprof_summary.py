"""Summarise a rocprofv3 --kernel-trace --stats run into a small text table:
    python scripts/rocprof_summary.py gpurun_out/prof_x/**/*_results.db [out.txt]      (rocpd sqlite output)
    python scripts/rocprof_summary.py gpurun_out/<tag>_stats [out.txt]                  (--output-format csv: a directory)
Per kernel name: calls, total ms, average us, share of GPU kernel time (+ VGPRs, LDS, grid from the database / the trace)."""
import csv
import glob
import os
import sqlite3
import sys


def main_csv(root, out=None):
    """rocprofv3 --output-format csv: <root>/**/*_kernel_stats.csv (+ *_kernel_trace.csv for registers / LDS / grid)."""
    stats = sorted(glob.glob(os.path.join(root, '**', '*_kernel_stats.csv'), recursive=True))
    if not stats:
        raise SystemExit('no *_kernel_stats.csv under ' + root)
    rows = list(csv.DictReader(open(stats[0])))
    extra = {}
    for path in glob.glob(os.path.join(root, '**', '*_kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(path)):
            name = r.get('Kernel_Name')
            if name and name not in extra:
                extra[name] = (r.get('VGPR_Count', ''), r.get('Accum_VGPR_Count', ''), r.get('LDS_Block_Size', ''),
                               r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '')))
    total = sum(float(r['TotalDurationNs']) for r in rows)
    lines = [f'# rocprofv3 --kernel-trace --stats summary of {stats[0]}', f'# total GPU kernel time {total / 1e6:.3f} ms',
             '# (registers per kernel: scripts/kernel_resources.py reads them from the code objects; the trace csv reports another unit)',
             f'{"calls":>7} {"total_ms":>10} {"avg_us":>10} {"min_us":>9} {"max_us":>9} {"share":>7} {"lds":>7} {"grid":>10} {"wg":>5}  name']
    for r in rows[:40]:
        vg, ag, lds, gx, wx = extra.get(r['Name'], ('', '', '', '', ''))
        lines.append(f'{int(r["Calls"]):7d} {float(r["TotalDurationNs"]) / 1e6:10.3f} {float(r["AverageNs"]) / 1e3:10.2f} '
                     f'{float(r["MinNs"]) / 1e3:9.2f} {float(r["MaxNs"]) / 1e3:9.2f} {float(r["Percentage"]):6.2f}% {lds:>7} '
                     f'{gx:>10} {wx:>5}  {r["Name"][:150]}')
    text = '\n'.join(lines)
    print(text)
    if out:
        open(out, 'w').write(text + '\n')


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    lines = [f'# rocprofv3 kernel-trace summary of {path}', f'# total GPU kernel time {total / 1e6:.3f} ms',
             f'{"calls":>7} {"total_ms":>10} {"avg_us":>10} {"min_us":>9} {"max_us":>9} {"share":>7} {"vgpr":>5} {"agpr":>5} {"lds":>7} {"grid":>9} {"wg":>5}  name']
    for name, calls, tot, avg, mn, mx, vg, ag, lds, gx, wx in rows[:60]:
        lines.append(f'{calls:7d} {tot / 1e6:10.3f} {avg / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} '
                     f'{100 * tot / total:6.2f}% {vg:5d} {ag:5d} {lds:7d} {gx:9d} {wx:5d}  {name[:150]}')
    text = '\n'.join(lines)
    print(text)
    if out:
        open(out, 'w').write(text + '\n')


if __name__ == '__main__':
    if os.path.isdir(sys.argv[1]):
        main_csv(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
    else:
        main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
