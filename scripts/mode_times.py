import sys
sys.path.insert(0,'/root/repo')
import __graft_entry__ as e
hg=e.load_package()
n,k=int(sys.argv[1]),int(sys.argv[2])
ctx=hg.Context(0); bfv=hg.BfvEncrypt.new(n,k); pk=bfv.setup(ctx)
w=hg.Witness.synthetic(bfv.params,3)
vals=hg.witness_gen(ctx,pk,w); out=hg.ProofBuffer()
for i in range(3):
    hg.prove_resident_mode(ctx,pk,vals,out,3); print(out.timings()["prove_ms"])
