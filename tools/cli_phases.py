import os, sys, time
t0 = time.time()
os.environ.setdefault('TEC_NO_TORCH', '1')
sys.path.insert(0, '.')
from te_counter_amd.measure import measureTE
t1 = time.time()
m = measureTE('x', 20)
m.bind_genome('/dev/shm/w.tecidx')
from te_counter_amd.index import FlatIndex
from te_counter_amd import Engine
m._engine = Engine(0)
t2 = time.time()
m.load_genome()
t3 = time.time()
r = m.parse_bampe('/dev/shm/w.bam')
t4 = time.time()
m.save_result_bulk(r, '/dev/shm/w.tsv')
t5 = time.time()
m._engine.close()
t6 = time.time()
print('imports %.3f  engine(hip init) %.3f  load_genome %.3f  parse %.3f  tsv %.3f  close %.3f' % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5))
