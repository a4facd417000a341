#!/bin/bash
# Before a gpurun visit that takes profiles: the commit the snapshot is made from (the GPU box has no .git).
#   tools/stamp_head.sh && gpurun -- 'bash tools/profile_round.sh r05_full full'
# "<hash>" for a clean tree, "<hash>+dirty" otherwise; bench.py and tools/traffic_json.py put it next to the csrc content hash.
cd "$(dirname "$0")/.." || exit 1
h=$(git rev-parse --short=12 HEAD)
git diff --quiet HEAD -- supersdr_amd include bench.py || h="$h+dirty"
echo "$h" > .ssdr_head
echo "$h"
