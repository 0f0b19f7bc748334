#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7) rocpd SQLite database: per-kernel stats and, when present,
PMC counter sums per kernel.  Usage: rocpd_summary.py results.db [out.md]"""
import sqlite3
import sys


def main():
  db = sys.argv[1]
  out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout
  con = sqlite3.connect(db)
  cur = con.cursor()
  cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
  name_col = 'name' if 'name' in cols else 'kernel_name'
  rows = cur.execute(
      f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
      f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
  total = sum(r[2] for r in rows) or 1
  print(f'# rocprofv3 kernel stats ({db})\n', file=out)
  print('| kernel | calls | total ms | avg us | min us | max us | % |', file=out)
  print('|---|---|---|---|---|---|---|', file=out)
  for n, c, tot, avg, mn, mx in rows:
    short = n if len(n) < 90 else n[:87] + '...'
    print(f'| `{short}` | {c} | {tot/1e6:.3f} | {avg/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {100*tot/total:.1f} |', file=out)
  try:
    pc = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    if pc:
      kn = 'kernel_name' if 'kernel_name' in pc else 'name'
      rows = cur.execute(
          f"select {kn}, counter_name, count(*), sum(value), avg(value) from counters_collection "
          f"group by {kn}, counter_name order by {kn}, counter_name").fetchall()
      if rows:
        print('\n# PMC counters (sum over dispatches; avg per dispatch)\n', file=out)
        print('| kernel | counter | dispatches | sum | avg/dispatch |', file=out)
        print('|---|---|---|---|---|', file=out)
        for n, cn, c, s, a in rows:
          short = n if len(n) < 60 else n[:57] + '...'
          print(f'| `{short}` | {cn} | {c} | {s:.4g} | {a:.4g} |', file=out)
  except sqlite3.Error as e:
    print(f'(no counters: {e})', file=out)


if __name__ == '__main__':
  main()
