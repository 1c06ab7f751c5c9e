"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite .db) into a small text table:
    python scripts/rocprof_summary.py gpurun_out/prof_x/**/*_results.db [skip_first_n_per_kernel]
Per kernel name: calls, total ms, average us, share of GPU kernel time, VGPRs, LDS."""
import sqlite3
import sys


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
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
