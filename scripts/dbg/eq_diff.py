"""Debug aid: proves one witness with the eq-factored PRODSUM form and (in a child process) without it, lists the proof sections that differ."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
n, k = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3:   # child: prove and dump
    import __graft_entry__ as g
    hg = g.load_package()
    ctx = hg.Context(0)
    bfv = hg.BfvEncrypt.new(n, k)
    pk = bfv.setup(ctx)
    w = hg.Witness.synthetic(bfv.params, 7)
    proof, tm = bfv.prove(ctx, pk, w)
    open(sys.argv[3], "wb").write(proof)
    sys.exit(0)
env = dict(os.environ)
env["HG_PROOF_MAP"] = "/tmp/map_eq.tsv"
subprocess.check_call([sys.executable, __file__, str(n), str(k), "/tmp/p_eq.bin"], env=env)
env["HG_NO_PS_EQ"] = "1"; env["HG_PROOF_MAP"] = "/tmp/map_ref.tsv"
subprocess.check_call([sys.executable, __file__, str(n), str(k), "/tmp/p_ref.bin"], env=env)
a = open("/tmp/p_eq.bin", "rb").read(); b = open("/tmp/p_ref.bin", "rb").read()
m = [(int(l.split("\t", 1)[0]), l.rstrip("\n").split("\t", 1)[1]) for l in open("/tmp/map_eq.tsv")]
print("lengths", len(a), len(b))
bad = 0
for (o, lab), (o2, _) in zip(m, m[1:] + [(len(a), "")]):
    if a[o:o2] != b[o:o2]:
        first = next(i for i in range(o, o2) if a[i] != b[i])
        print("DIFF  %-110s first differing element %d of %d" % (lab[:110], (first - o) // 16, (o2 - o) // 16))
        for q in range(min(8, (o2 - o) // 16)):
            print("   ", q, a[o + 16 * q:o + 16 * q + 16].hex(), b[o + 16 * q:o + 16 * q + 16].hex(), "" if a[o + 16 * q:o + 16 * q + 16] == b[o + 16 * q:o + 16 * q + 16] else "<--")
        bad += 1
        if bad > 40: break
print("sections differing:", bad)
