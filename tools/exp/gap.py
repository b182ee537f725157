import csv, sys, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted([r for r in rows if 'k_stream' in r['Kernel_Name']], key=lambda r: int(r['Start_Timestamp']))[-40:]
dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in ks]
gap = [(int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3 for a, b in zip(ks[:-1], ks[1:])]
per = [(int(b['Start_Timestamp']) - int(a['Start_Timestamp'])) / 1e3 for a, b in zip(ks[:-1], ks[1:])]
print('k_stream dur us median %.1f  gap median %.1f  period %.1f' % (st.median(dur), st.median(gap), st.median(per)))
