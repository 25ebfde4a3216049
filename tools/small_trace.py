import csv,glob,os,sys
f=sorted(glob.glob(sys.argv[1]+'/*/*kernel_trace.csv'), key=os.path.getmtime)[-1]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# take a window in the middle of the first config: find diag128 kernels; steps delimited by diag128 with factor
diag=[i for i,r in enumerate(rows) if 'diag128' in r['Kernel_Name']]
# first config has 30 steps, each with >=1 diag (forward) ... pick steps 10..11
a,b=diag[20],diag[22]
t0=int(rows[a]['Start_Timestamp']); prev=t0; busy=0
for r in rows[a:b]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print(f"{(s-t0)/1e3:8.1f} +{(e-s)/1e3:6.1f} gap {(s-prev)/1e3:6.1f}  {r['Kernel_Name'][:70]}")
    busy+=e-s; prev=e
print("wall",(prev-t0)/1e3,"busy",busy/1e3,"kernels",b-a)
