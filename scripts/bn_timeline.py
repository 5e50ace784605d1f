import csv, sys, collections
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("hg::bn::","").replace("void ",""), r.get("Queue_Id","?")) for r in csv.DictReader(open(sys.argv[1]))), key=lambda r: r[0])
# last prove: from the last k_bn_low_limb backwards to the preceding k_bn_lift_signed run start
idx = max(i for i,r in enumerate(rows) if "k_bn_low_limb" in r[2])
# find start of that prove: first kernel after the last k_bn_gate_eval before idx
gi = max(i for i,r in enumerate(rows[:idx]) if "k_bn_gate_eval" in r[2])
last = rows[gi+1:]
t0 = last[0][0]
print("kernels", len(last), "span %.2f ms" % ((max(e for _,e,_,_ in last)-t0)/1e6))
B = 500000
buckets = collections.defaultdict(lambda: collections.defaultdict(float))
names = collections.defaultdict(lambda: collections.defaultdict(float))
for s,e,n,q in last:
    b = (s - t0)//B
    buckets[b][q] += (e-s)/1e3
    names[b][n] += (e-s)/1e3
for b in sorted(buckets):
    top = sorted(names[b].items(), key=lambda kv:-kv[1])[:3]
    print("%5.1f ms: %s | %s" % (b*0.5, " ".join("q%s=%4.0fus" % (q, v) for q,v in sorted(buckets[b].items())), ", ".join("%s %.0f" % (k[:28],v) for k,v in top)))
