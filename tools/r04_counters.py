#!/usr/bin/env python
"""Folds the rocprofv3 --pmc passes of tools/r04_counters.sh (gpurun_out/pmc_r04_*/) into one JSON: per (layer, kernel) the raw SQ
counters (averaged over dispatches) and their shares of SQ_WAVE_CYCLES (issuing / issue-stalled / parked; LDS issue stall;
MFMA-busy cycles per CU-cycle).    python tools/r04_counters.py > gpurun_out/r04_sq_counters.json"""
import collections
import csv
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for d in sorted(glob.glob(os.path.join(ROOT, 'gpurun_out', 'pmc_r04_*'))):
    tag = os.path.basename(d)[len('pmc_r04_'):]
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for fn in glob.glob(os.path.join(d, '*', '*counter_collection.csv')):
        for r in csv.DictReader(open(fn)):
            k = r['Kernel_Name']
            if 'conv' not in k:
                continue
            k = re.sub(r'void \(anonymous namespace\)::', '', k)[:70]
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
    log = ''
    try:
        log = open(os.path.join(d, 'run.log')).read().strip().splitlines()[-1]
    except Exception:                                  # noqa
        pass
    rec = {'run': log, 'kernels': {}}
    for k, v in agg.items():
        c = {n: x / cnt[(k, n)] for n, x in v.items()}
        wc = c.get('SQ_WAVE_CYCLES', 0) or 1.0
        c = {n: round(x) for n, x in c.items()}
        c['share_of_wave_cycles'] = {'issuing': round(c.get('SQ_ACTIVE_INST_ANY', 0) / wc, 3), 'issue_stalled': round(c.get('SQ_WAIT_INST_ANY', 0) / wc, 3),
                                     'parked': round(c.get('SQ_WAIT_ANY', 0) / wc, 3), 'lds_issue_stall': round(c.get('SQ_WAIT_INST_LDS', 0) / wc, 3)}
        rec['kernels'][k] = c
    out[tag] = rec
print(json.dumps(out, indent=1))
