import csv, glob, sys, collections
def load(d):
    f = glob.glob(d + '/runc/*_counter_collection.csv')[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name']].append(float(r['Counter_Value']))
    return acc
for tag in sys.argv[1:]:
    fe, wr = load(f'pmc_{tag}_FETCH_SIZE'), load(f'pmc_{tag}_WRITE_SIZE')
    print(f'== {tag}: per-launch averages (KB as reported; FETCH x2 corrected for 16B/lane streaming reads)')
    for k in sorted(fe, key=lambda k: -sum(fe[k])):
        f = sum(fe[k]) / len(fe[k]); w = sum(wr.get(k, [0])) / max(1, len(wr.get(k, [0])))
        print(f'{k[:60]:60s} n={len(fe[k]):4d} FETCH={f/1024:9.2f} MB (x2={2*f/1024:9.2f}) WRITE={w/1024:9.2f} MB')
