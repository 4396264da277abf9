#!/bin/bash
# Both payloads, one setting each (development aid): bash tools/ab_two.sh <illumina records> <random records> [ENV...]
cd $GRAFT_REPO_ROOT
NI=$1; NR=$2; shift 2
for P in illumina random; do
  N=$NI; [ $P = random ] && N=$NR
  bash tools/ab_env.sh $N $P "$*" 2>&1 | grep -v amdgpu.ids | sed "s/^/$P /"
done
