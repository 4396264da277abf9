import sys, os, cProfile, pstats
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import bench_e2e
from te_counter_amd import synth
ann = synth.make_annotation('hg38', scale=0.2)
pr = cProfile.Profile()
pr.enable()
out = bench_e2e.run_sc(10_000_000, ann, reps=2)
pr.disable()
print(out)
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
