"""Goldilocks prove times around an event that drops the cached launch graph (one_stream / profile1 / profile2 / graph_off / bn):
with stream priorities on (HG_PRIO=1) every re-captured graph replays at ~5 ms instead of ~3 ms."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as entry
hg = entry.load_package()
n, k = 32768, 16
ctx = hg.Context(0)
bfv = hg.BfvEncrypt.new(n, k)
pk = bfv.setup(ctx)
w = hg.Witness.synthetic(bfv.params, 0x4752454330 + n)
vals = hg.witness_gen(ctx, pk, w)
out = hg.ProofBuffer()
def step(tag):
    hg.prove_resident(ctx, pk, vals, out); print(tag, "%.3f" % out.timings()["gpu_ms"], "%.3f" % out.timings()["prove_ms"])
for i in range(5): step("fresh %d" % i)
mode = sys.argv[1]
if mode == "one_stream":
    ctx.set_option("one_stream", 1); step("one_stream"); ctx.set_option("one_stream", 0)
elif mode == "profile1":
    ctx.profile_select("sc_round2<grand_product,ext>"); ctx.profile(1); step("prof1"); step("prof1"); ctx.profile(0)
elif mode == "profile2":
    ctx.profile(2); step("prof2"); ctx.profile(0)
elif mode == "bn":
    t = ctx.prove_bn254(pk, w, cap=1 << 25); t = ctx.prove_bn254(pk, w, cap=1 << 25); print("bn254 prove ms", t[1:])
elif mode == "graph_off":
    ctx.set_option("graph", 0); step("nograph"); ctx.set_option("graph", 1)
for i in range(6): step("after %d" % i)
