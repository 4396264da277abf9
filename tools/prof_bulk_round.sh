#!/bin/bash
# The bulk half of tools/prof_round.sh (run on the GPU box): the five count-kernel workloads bench.py reports a roofline for.
#   bash tools/prof_bulk_round.sh <round tag, e.g. r03>
R=${1:-r03}
cd $GRAFT_REPO_ROOT
HBM_ONLY=1 bash tools/prof.sh ${R}_headline || exit 1
HBM_ONLY=1 bash tools/prof.sh ${R}_headline_sorted --order sorted || exit 1
HBM_ONLY=1 bash tools/prof.sh ${R}_cfg2_shuffled --layout se --records 100000000 || exit 1
HBM_ONLY=1 bash tools/prof.sh ${R}_cfg2_sorted --layout se --order sorted --records 100000000 || exit 1
HBM_ONLY=1 bash tools/prof.sh ${R}_cfg5 --genome custom3m --layout se --records 100000000 || exit 1
