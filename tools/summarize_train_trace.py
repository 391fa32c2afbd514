import os
import csv,glob,re,sys
f=max(glob.glob(sys.argv[1]+"/*/*kernel_stats.csv"), key=os.path.getmtime)
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows); n=sum(int(r["Calls"]) for r in rows)
print(n, round(tot/1e6,2))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:24]:
    name=re.sub(r"\(.*","",r["Name"].replace("(anonymous namespace)::","").replace("void ",""))[:60]
    print(name, r["Calls"], round(float(r["AverageNs"])/1e3,1), round(float(r["TotalDurationNs"])/tot*100,1))
