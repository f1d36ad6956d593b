#!/bin/bash
# Boxes of the pool differ by ~7 % on the same binary.  Same-box A/B of the occupancy conv1 first (kernel-map path, then
# default), then the round's artefacts -- only if the box is of the fast kind, so that profiles/ stays comparable with
# the earlier rounds' (which came from fast boxes): THRESH pairs/s on the kernel-map path.
set -o pipefail
TAG=${1:-r03d}
THRESH=${2:-2280}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
val() { python3 -c "import sys, json; print(json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])['value'])" $1; }
APR_OCC_CONV=0 python3 $R/bench.py --steps 200 --no-workloads --no-cpu-baseline --no-roofline > $R/gpurun_out/ab_occ_off.json 2> $R/gpurun_out/ab_occ_off.err </dev/null || exit 1
python3 $R/bench.py --steps 200 --no-workloads --no-cpu-baseline --no-roofline > $R/gpurun_out/ab_occ_on.json 2> $R/gpurun_out/ab_occ_on.err </dev/null || exit 1
OFF=$(val $R/gpurun_out/ab_occ_off.json); ON=$(val $R/gpurun_out/ab_occ_on.json)
echo "[ab] kernel-map conv1 $OFF pairs/s, occupancy conv1 $ON pairs/s" | tee $R/gpurun_out/ab_occ.txt
if python3 -c "import sys; sys.exit(0 if float('$OFF') >= float('$THRESH') else 1)"; then
  bash $R/scripts/profile_round.sh $TAG > $R/gpurun_out/profile_$TAG.log 2>&1 || { tail -5 $R/gpurun_out/profile_$TAG.log; exit 1; }
  echo "[ab] profiled as $TAG"
else
  echo "[ab] slow box: not profiled"
fi
