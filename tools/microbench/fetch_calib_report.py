import csv,glob,sys
agg={}
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=(r['Kernel_Name'].split('(')[0],r['Dispatch_Id'],r['Counter_Name'])
        agg[k]=agg.get(k,0)+float(r['Counter_Value'])
for k,v in sorted(agg.items()): print("%-12s dispatch %-3s %-12s %12.0f KiB = %.3f of 1 GiB"%(k[0],k[1],k[2],v,v/1048576))
