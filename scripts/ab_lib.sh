#!/bin/bash
# same-box A/B of two builds of libapr_hip.so on the default bench: ab_lib.sh OLD.so NEW.so [reps] [extra bench args]
# (build the old tree with `git stash; python -m apr_amd.build; cp apr_amd/lib/libapr_hip.so /somewhere/old.so; git stash pop`,
# keep both under apr_amd/lib/ so that they travel with the snapshot)
set -euo pipefail
OLD=$1; NEW=$2; REPS=${3:-3}; shift $(( $# < 3 ? $# : 3 ))
L=$(cd "$(dirname "$0")/.." && pwd)/apr_amd/lib/libapr_hip.so
# every install is copy + rename: a NEW inode each time, so a process that still has the previous build mapped keeps a whole
# library (cp over the live file truncates and rewrites the mapped inode)
install_lib() { cp "$1" "$L.tmp" && mv -f "$L.tmp" "$L"; }
cp "$L" "$L.keep"
trap 'install_lib "$L.keep"; rm -f "$L.keep"' EXIT
for i in $(seq "$REPS"); do
  for v in old new; do
    if [ $v = old ]; then install_lib "$OLD"; else install_lib "$NEW"; fi
    r=$(timeout -k 10 300 python "$(dirname "$0")/../bench.py" --no-cpu-baseline --no-roofline --no-workloads "$@" 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],1))")
    echo "$v $r"
  done
done
