#!/bin/bash
# gpurun with the commit of the tree it sends left in gpurun_in/HEAD (the GPU box's snapshot has no .git; tools/pmc_counters.py
# reads it for `collected_on`).  usage: bash tools/gpurun.sh [--timeout S] -- '<command>'
mkdir -p "$(dirname "$0")/../gpurun_in"
git -C "$(dirname "$0")/.." rev-parse --short HEAD > "$(dirname "$0")/../gpurun_in/HEAD"
git -C "$(dirname "$0")/.." diff --quiet || echo "+uncommitted" >> "$(dirname "$0")/../gpurun_in/HEAD"
exec /usr/local/graft/bin/gpurun "$@"
