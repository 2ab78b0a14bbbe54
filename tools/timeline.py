import csv,glob,sys
d=sys.argv[1]
ev=[]
for f in glob.glob(d+'/**/*memory_copy_trace.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Direction'] if 'Direction' in r else r.get('Name','copy'),r.get('Stream_Id','?'), r.get('Bytes', r.get('Size','?'))))
for f in glob.glob(d+'/**/*kernel_trace.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:40],r.get('Stream_Id','?'),''))
ev.sort()
t0=ev[0][0]
# print the last 150 events
for s,e,n,st,b in ev[-160:]:
    print("%10.3f %10.3f %8.3f ms  s%-3s %-42s %s"%((s-t0)/1e6,(e-t0)/1e6,(e-s)/1e6,st,n,b))
