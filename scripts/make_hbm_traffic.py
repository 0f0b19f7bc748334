#!/usr/bin/env python
"""profiles/hbm_traffic.json from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass).

  make_hbm_traffic.py fetch.db write.db out.json "source note"

HBM bytes per launch = avg FETCH_SIZE x 1024 x 2 (MI355X_MICROARCH.md "HBM": gfx950 tallies the 128-B requests of
16-B/lane streaming reads at 64 B) + avg WRITE_SIZE x 1024 (uncorrected).  The file records the hash of the kernel sources
(bench.kernel_source_sha) so bench.py reports `traffic: null` once the kernels have changed."""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_kernel(db, counter):
  cur = sqlite3.connect(db).cursor()
  pc = [r[1] for r in cur.execute('pragma table_info(counters_collection)')]
  kn = 'kernel_name' if 'kernel_name' in pc else 'name'
  rows = cur.execute(f"select {kn}, avg(value) from counters_collection where counter_name = ? group by {kn}", (counter,)).fetchall()
  return {n.split('(')[0].replace('void ', ''): a for n, a in rows}


def main():
  fetch_db, write_db, out, note = sys.argv[1:5]
  from bench import kernel_source_sha
  f, w = per_kernel(fetch_db, 'FETCH_SIZE'), per_kernel(write_db, 'WRITE_SIZE')
  kernels = {k: {'fetch_bytes': 2.0 * 1024.0 * f[k], 'write_bytes': 1024.0 * w.get(k, 0.0)} for k in sorted(f)}
  json.dump({'note': 'HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KB units); FETCH doubled '
                     'per MI355X_MICROARCH.md (gfx950 tallies 128-B requests of 16-B/lane streaming reads at 64 B); WRITE '
                     'uncorrected.', 'source': note, 'csrc_sha16': kernel_source_sha(), 'kernels': kernels},
            open(out, 'w'), indent=1)
  print(out, len(kernels), 'kernels')


if __name__ == '__main__':
  main()
