#!/bin/bash
# The round's traffic passes in one go (run on the GPU box): the five bulk workloads bench.py reports a roofline for
# (HBM passes only) and the single-cell job.  bash tools/prof_round.sh <round tag, e.g. r03>
R=${1:-r03}
cd $GRAFT_REPO_ROOT
HBM_ONLY=1 bash tools/prof.sh ${R}_headline || exit 1
HBM_ONLY=1 bash tools/prof.sh ${R}_headline_sorted --order sorted || exit 1
HBM_ONLY=1 bash tools/prof.sh ${R}_cfg2_shuffled --layout se --records 100000000 || exit 1
HBM_ONLY=1 bash tools/prof.sh ${R}_cfg2_sorted --layout se --order sorted --records 100000000 || exit 1
HBM_ONLY=1 bash tools/prof.sh ${R}_cfg5 --genome custom3m --layout se --records 100000000 || exit 1
bash tools/prof_sc.sh ${R}_sc_cfg4 || exit 1
