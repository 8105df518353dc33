import csv, glob, sys, os
rows=[]
for p in glob.glob(os.path.join(sys.argv[1],"**","*kernel_trace.csv"),recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]))
rows.sort()
fused=[r for r in rows if "k_propagate_classify" in r[2]]
# group into frames: a frame starts at a kernel whose template args have TOP=true (5th arg)
frames=[]
for s,e,k in fused:
    args=k[k.index("<")+1:k.index(">")].split(", ")
    if args[4]=="true": frames.append([])
    if frames: frames[-1].append((s,e))
prev=None
for i,f in enumerate(frames):
    per=(f[0][0]-prev)/1e3 if prev else 0
    prev=f[0][0]
    print(i, "period %.1f"%per, " ".join("%.1f"%((e-s)/1e3) for s,e in f))
