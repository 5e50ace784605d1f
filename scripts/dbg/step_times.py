import sys, time
sys.path.insert(0,'/root/repo')
import __graft_entry__ as e
hg=e.load_package()
ctx=hg.Context(0); bfv=hg.BfvEncrypt.new(32768,16); pk=bfv.setup(ctx)
w=hg.Witness.synthetic(bfv.params,3)
vals=hg.witness_gen(ctx,pk,w); out=hg.ProofBuffer()
for i in range(12):
    t0=time.perf_counter(); hg.prove_resident(ctx,pk,vals,out); dt=(time.perf_counter()-t0)*1e3
    t=out.timings()
    if i>=6: print("call %.3f ms: %s" % (dt, {k: round(v,3) for k,v in t.items()}))
