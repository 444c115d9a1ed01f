"""Timeline of the last few-candidates call in a rocprofv3 kernel trace (tools/few_trace.sh): python tools/few_timeline.py trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'kstar_rows' in r['Kernel_Name']]
i0 = idx[-1]
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i0 + int(sys.argv[2]) if len(sys.argv) > 2 else None]:
    n = r['Kernel_Name'].split('(')[0][-44:]
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:8.1f} {(int(r['End_Timestamp'])-t0)/1e3:8.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:6.1f} q{r['Queue_Id']} {n} grid {r['Grid_Size_X']}x{r['Grid_Size_Y']} vgpr {r['VGPR_Count']} lds {r['LDS_Block_Size']} scr {r['Scratch_Size']}")
