import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_stem' in r['Kernel_Name']]
a=idx[40]; b=idx[41]
t0=int(rows[a-3]['Start_Timestamp'])
for r in rows[a-3:b-3]:
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:8.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f} q{r['Queue_Id']} {r['Kernel_Name'][:50]} grid={r['Grid_Size_X']}x{r['Grid_Size_Y']} v{r['VGPR_Count']}")
