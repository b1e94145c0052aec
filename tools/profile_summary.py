#!/usr/bin/env python
"""Turns rocprofv3 outputs into the committed summaries under profiles/:
   python tools/profile_summary.py <stats_dir> <fetch_dir> <write_dir> <round-tag>
 - <tag>_kernel_stats.csv : copy of the --kernel-trace --stats table (per-kernel calls / average ns)
 - <tag>_pmc_summary.json (+ pmc_summary.json for a round's headline tag rNN): per kernel instance (bench.py naming) HBM bytes per launch from the TCC counters,
                            corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE is in KiB and reports HALF of a wide
                            coalesced read stream on gfx950 (doubled here); WRITE_SIZE in KiB is exact."""
import csv, glob, json, os, re, shutil, sys, collections


def pretty(name):
    """rocprof kernel name -> bench.py naming, e.g. conv_fwd_kernel<bf16,8,16,64,4,1,3,3,1>.  The kernels take the dtype
    as an int template argument (0 = f32, 1 = bf16) so that the names demangle cleanly."""
    m2 = re.search(r'(wgrad_sweep_kernel|conv_sweep_kernel|conv_first_win_kernel|conv_first_mfma_kernel)<([^>]*)>', name)
    if m2:          # round-3 kernels: integer / bool template arguments only, kept as rocprof prints them (bench.py's naming)
        return '%s<%s>' % (m2.group(1), ','.join(a.strip() for a in m2.group(2).split(',')))
    m = re.search(r'(conv_fwd_glds_kernel|conv_fwd_kernel|conv_wgrad_kernel)<([^>]*)>', name)
    if not m:
        mm = re.search(r'(conv_fwd_glds_kernel|conv_fwd_kernel|conv_wgrad_kernel)I((?:Li\d+E)+)', name)
        if mm:
            m = re.match(r'(.*)', mm.group(1)); args = re.findall(r'Li(\d+)E', mm.group(2)); kname = mm.group(1)
        else:
            k = re.search(r'(\w+_kernel)', name)
            return re.sub(r'^_ZN\d+_GLOBAL__N_1\d+', '', k.group(1)) if k else name[:48]
    else:
        kname = m.group(1); args = [a.strip() for a in m.group(2).split(',')]
        if kname == 'conv_fwd_kernel' and len(args) == 10:      # trailing POOL flag: same family as the plain instance
            args = args[:9]
    if kname != 'conv_fwd_glds_kernel':
        args[0] = {'0': 'f32', '1': 'bf16'}.get(args[0], args[0])
    return '%s<%s>' % (kname, ','.join(args))


def pmc(dirname, counter):
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for fn in glob.glob(os.path.join(dirname, '*', '*counter_collection.csv')):
        for r in csv.DictReader(open(fn)):
            if r['Counter_Name'] == counter:
                k = pretty(r['Kernel_Name']); agg[k] += float(r['Counter_Value']); cnt[k] += 1
    return {k: agg[k] / cnt[k] for k in agg}


def main():
    stats_dir, fetch_dir, write_dir, tag = sys.argv[1:5]
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles')
    os.makedirs(root, exist_ok=True)
    st = glob.glob(os.path.join(stats_dir, '*', '*kernel_stats.csv'))[0]
    rows = list(csv.DictReader(open(st)))
    merged = collections.OrderedDict()          # template instances that differ only in a dropped flag share a row
    for r in rows:
        a = merged.setdefault(pretty(r['Name']), [0, 0.0, 0.0])
        a[0] += int(r['Calls']); a[1] += float(r['TotalDurationNs']); a[2] += float(r['Percentage'])
    with open(os.path.join(root, tag + '_kernel_stats.csv'), 'w') as f:
        w = csv.writer(f); w.writerow(['kernel', 'calls', 'total_ms', 'avg_us', 'percent'])
        for k, (calls, tot, pct) in sorted(merged.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, calls, '%.3f' % (tot / 1e6), '%.2f' % (tot / calls / 1e3), '%.4g' % pct])
    fetch, write = pmc(fetch_dir, 'FETCH_SIZE'), pmc(write_dir, 'WRITE_SIZE')
    out = {}
    for k in sorted(set(fetch) | set(write)):
        fb, wb = fetch.get(k, 0.0) * 1024 * 2, write.get(k, 0.0) * 1024
        out[k] = {'hbm_bytes_per_launch': int(fb + wb), 'fetch_bytes_corrected_x2': int(fb), 'write_bytes': int(wb)}
    out['_commit'] = tag                        # which build / round the counters belong to (bench.py prints it as traffic_source)
    json.dump(out, open(os.path.join(root, tag + '_pmc_summary.json'), 'w'), indent=1, sort_keys=True)
    if re.fullmatch(r'r\d+', tag):               # the headline workload of a round: what bench.py's roofline.traffic reads
        json.dump(out, open(os.path.join(root, 'pmc_summary.json'), 'w'), indent=1, sort_keys=True)
    print('wrote', tag + '_kernel_stats.csv', 'and', tag + '_pmc_summary.json with', len(out) - 1, 'kernels')


if __name__ == '__main__':
    main()
