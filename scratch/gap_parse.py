import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted([(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:40], r.get('Grid_Size', r.get('Grid_Size_X', ''))) for r in csv.DictReader(open(f))])
gaps = collections.defaultdict(list)
for a, b in zip(rows[:-1], rows[1:]):
  if a[2] == b[2] and a[3] == b[3]: gaps[(a[2], a[3])].append((b[0] - a[1]) / 1e3)
for k, v in gaps.items(): print(k, 'dur_us', round((rows[0][1]-rows[0][0])/1e3,1), 'gaps_us', [round(x, 1) for x in v[:12]])
