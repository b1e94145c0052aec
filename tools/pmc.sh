#!/bin/bash
# usage: tools/pmc.sh <tag> <counters...> -- <conv_micro args>   (run on the GPU box)
tag=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/${PMC_TOOL:-conv_micro.py} "$@" > $out/run.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$out/*/*counter_collection.csv")
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r['Kernel_Name'][:60]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
for k,v in agg.items():
    if 'conv' in k or 'head' in k or 'pool' in k:
        print(k, {c: round(x/cnt[(k,c)],1) for c,x in v.items()})
PY
tail -1 $out/run.log
