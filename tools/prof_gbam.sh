#!/bin/bash
# Profiles the on-GPU BAM decoder (run on the GPU box through gpurun): bash tools/prof_gbam.sh [records]
# kernel trace + two PMC passes over a synthetic PE BAM, both payloads; results under gpurun_out/prof_gbam_*/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
N=${1:-3000000}
python3 - <<PY
import sys, os
sys.path.insert(0, '.')
from te_counter_amd import bam, synth
ann = synth.make_annotation('hg38', scale=0.05)
u = synth.make_units(ann, $N, seed=4242, paired=True)
refs = ann.chrom_names + ['chrUn_1']
tid, pos, fl, mq = bam.units_to_records(u, len(refs) - 1, True)
for p in ('random', 'illumina'):
    bam.write_bam_arrays('/tmp/prof_%s.bam' % p, refs, tid, pos, fl, mq, pair_names=True, payload=p)
ann.flat.save('/tmp/prof.tecidx')
PY
cat > /tmp/run_one.py <<'PY'
import sys, time
sys.path.insert(0, '.')
from te_counter_amd.measure import measureTE
m = measureTE('x', 20); m.bind_genome('/tmp/prof.tecidx'); m.load_genome()
for rep in range(2):
    t = time.time(); r = m.parse_bampe(sys.argv[1]); print('decode_and_count', time.time() - t, m.decode_report['seconds'])
PY
for p in random illumina; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_gbam_$p/trace -o x -- python3 /tmp/run_one.py /tmp/prof_$p.bam > gpurun_out/prof_gbam_$p.log 2>&1
  grep decode_and_count gpurun_out/prof_gbam_$p.log
  [ -n "$TRACE_ONLY" ] && continue
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/prof_gbam_$p/pmc1 -o x -- python3 /tmp/run_one.py /tmp/prof_$p.bam > gpurun_out/prof_gbam_${p}_pmc1.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_IFETCH --output-format csv -d gpurun_out/prof_gbam_$p/pmc2 -o x -- python3 /tmp/run_one.py /tmp/prof_$p.bam > gpurun_out/prof_gbam_${p}_pmc2.log 2>&1
done
find gpurun_out/prof_gbam_random -name "*.csv" | head
