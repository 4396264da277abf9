#!/bin/bash
# Kernel trace and PMC passes of the on-GPU BAM decoder on one synthetic PE BAM with FULL-SIZE inflate launches
# (run on the GPU box): bash tools/prof_gbam2.sh <records> <payload> <tag> [passes: trace pmc1 pmc2 fetch write]
# Results: gpurun_out/<tag>/{trace,pmc1,...}/ and gpurun_out/<tag>_summary.txt (per-kernel means of every counter).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
N=$1; P=$2; TAG=$3; shift 3
PASSES=${*:-trace pmc1 pmc2}
python3 - <<PY
import sys; sys.path.insert(0, '.')
from te_counter_amd import bam, synth
ann = synth.make_annotation('hg38', scale=0.05)
u = synth.make_units(ann, $N // 2, seed=4242, paired=True)
refs = ann.chrom_names + ['chrUn_1']
tid, pos, fl, mq = bam.units_to_records(u, len(refs) - 1, True)
bam.write_bam_arrays('/dev/shm/prof.bam', refs, tid, pos, fl, mq, pair_names=True, payload='$P')
ann.flat.save('/dev/shm/prof.tecidx')
PY
cat > /tmp/prof_one.py <<'PY'
import sys, time
sys.path.insert(0, '.')
from te_counter_amd.measure import measureTE
m = measureTE('x', 20); m.bind_genome('/dev/shm/prof.tecidx'); m.load_genome()
for rep in range(2):
    t = time.time(); r = m.parse_bampe('/dev/shm/prof.bam'); print('decode_and_count %.3f' % (time.time() - t), m.decode_report['blocks'], m.decode_report['seconds'], flush=True)
PY
mkdir -p gpurun_out/$TAG
for pass in $PASSES; do
  case $pass in
    trace) FLAGS="--kernel-trace --stats" ;;
    pmc1) FLAGS="--pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" ;;
    pmc2) FLAGS="--pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES" ;;
    fetch) FLAGS="--pmc FETCH_SIZE" ;;
    write) FLAGS="--pmc WRITE_SIZE" ;;
  esac
  timeout -s KILL 120 rocprofv3 $FLAGS --output-format csv -d gpurun_out/$TAG/$pass -o x -- python3 /tmp/prof_one.py > gpurun_out/$TAG/$pass.log 2>&1 || { echo "pass $pass failed"; tail -5 gpurun_out/$TAG/$pass.log; break; }
  grep decode_and_count gpurun_out/$TAG/$pass.log | sed "s/^/$pass /"
done
rm -f /dev/shm/prof.bam /dev/shm/prof.tecidx
python3 - <<PY
import csv, glob, collections
out = open('gpurun_out/$TAG' + '_summary.txt', 'w')
for f in glob.glob('gpurun_out/$TAG/trace/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print('%-70s calls %4s avg %10.3f ms min %10.3f max %10.3f' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e6, float(r['MinNs']) / 1e6, float(r['MaxNs']) / 1e6), file=out)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/$TAG/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'ginflate' in k or 'gresolve' in k:
            # only the full-size launches (grid >= 60000 blocks of phase 1, the same dispatches of phase 2)
            agg[('P1 ' if 'ginflate' in k else 'P2 ') + r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    print(k, file=out)
    for c, v in sorted(agg[k].items()):
        print('   %-28s n=%d mean %.4g' % (c, len(v), sum(v) / len(v)), file=out)
out.close()
print(open('gpurun_out/$TAG' + '_summary.txt').read())
PY
