import os, sys
sys.path.insert(0, os.getcwd())
import __graft_entry__ as entry
hg = entry.load_package()
ctx = hg.Context(0); bfv = hg.BfvEncrypt.new(32768, 16); pk = bfv.setup(ctx)
w = hg.Witness.synthetic(bfv.params, 0x4752454330 + 5)
for i in range(4):
    sys.stderr.write("=== prove %d\n" % i)
    proof, wms, pms = ctx.prove_bn254(pk, w, cap=1 << 25)
    sys.stderr.write("prove_ms %.3f witness %.3f\n" % (pms, wms))
