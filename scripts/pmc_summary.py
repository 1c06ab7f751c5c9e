"""Collapse rocprofv3 --pmc csv output (one directory per pass) into per-kernel averages."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root):
    agg = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row.get('Kernel_Name', '')
                if 'bp::' not in name:
                    continue
                short = name.split('(')[0].replace('void ', '')
                agg[short][row['Counter_Name']].append(float(row['Counter_Value']))
    for kern, counters in sorted(agg.items()):
        print(kern)
        for c, vals in sorted(counters.items()):
            print(f'    {c:32s} n={len(vals):4d} avg={sum(vals) / len(vals):18.1f}')


if __name__ == '__main__':
    main(sys.argv[1])
