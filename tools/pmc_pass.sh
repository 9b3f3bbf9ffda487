#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <counters...> -- <command...>   (one rocprofv3 PMC pass; summary -> gpurun_out/pmc_<tag>.txt)
# Counters are collected in their own pass with --kernel-trace only (no sys/hip traces), as the pool requires.
tag=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
out=/tmp/pmc_$tag
rm -rf $out
( cd $GRAFT_REPO_ROOT && timeout 150 rocprofv3 --kernel-trace --pmc "${ctrs[@]}" -d $out -o p --output-format csv -- "$@" > /tmp/pmc_$tag.log 2>&1 )
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 - "$f" > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); cnt[k] += 1
for k, n in cnt.most_common():
    print(f"{k}  launches={n}")
    for c, v in sorted(agg[k].items()):
        print(f"    {c:32s} {v / n:16.1f} per launch")
PY
cat $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.txt | head -60
