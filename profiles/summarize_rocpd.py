#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average.

usage: summarize_rocpd.py <results.db> [--all]   (default: only this repo's asl:: kernels
plus the 5 largest foreign kernels, names shortened)"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*', '', name)
    name = name.replace('void ', '')
    return name if len(name) < 90 else name[:87] + '...'


def main():
    db = sys.argv[1]
    show_all = '--all' in sys.argv
    c = sqlite3.connect(db)
    rows = list(c.execute('select name, total_calls, total_duration, average, percentage '
                          'from top_kernels'))
    ours = [r for r in rows if 'asl::' in r[0]]
    other = [r for r in rows if 'asl::' not in r[0]]
    print(f'{"kernel":<60} {"calls":>7} {"total_ms":>12} {"avg_us":>12} {"pct":>7}')
    for r in ours + (other if show_all else other[:5]):
        print(f'{short(r[0]):<60} {r[1]:>7} {r[2] / 1e3:>12.3f} {r[3]:>12.3f} {r[4]:>7.2f}')
    print(f'# {len(rows)} distinct kernels; durations from rocprofv3 --kernel-trace (us)')


if __name__ == '__main__':
    main()
