import sys, time
from torchain_amd import io, synth
from torchain_amd._lib import lib
for k in sys.argv[2:]:
    lib.tc_debug_set(k.encode(), 1)
lib.tc_debug_set(b"sched_trace", 1)
name = sys.argv[1]
fst = synth.config_den_fst(name)
t=time.time()
gr = io.DenominatorGraph(fst, fst.num_pdfs)
st = gr.stats()
print(name, st['fwd_conflict_x1000'], st['bwd_conflict_x1000'], st['tied'], "%.2fs" % (time.time()-t))
