"""Print the headline of bench.py JSON lines: value, ms_per_step and the roofline rows whose name matches argv[2]."""
import json
import sys

for path in sys.argv[1].split(','):
    d = json.loads(open(path).read().strip().splitlines()[-1])
    pat = sys.argv[2] if len(sys.argv) > 2 else None
    rows = [(l.get('op'), l.get('us')) for l in (d.get('roofline') or {}).get('layers', [])
            if pat and pat in str(l.get('op'))]
    print(d['config']['workload'][:40], d['value'], d['ms_per_step'], rows)
