"""Per-level kernel timeline of the last TSVQ build in a rocprofv3 kernel trace (csv); `--sum` prints totals per kernel name."""
import collections, csv, re, sys
tr = list(csv.DictReader(open(sys.argv[1])))
ks = []
for r in tr:
    m = re.search(r'(k_\w+)(<[^>]*>)?', r['Kernel_Name'])
    if m:
        ks.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), m.group(0)))
ks.sort()
idx = max(i for i, k in enumerate(ks) if k[2].startswith(('k_build_init', 'k_iota')))
sel = ks[idx:]
t0 = sel[0][0]
args = sys.argv[2:]
if args and args[0] == '--sum':
    tot = collections.Counter(); cnt = collections.Counter()
    for s, e, n in sel:
        tot[n] += e - s; cnt[n] += 1
    for n, v in tot.most_common():
        print(f"{v / 1e3:9.1f} us  {cnt[n]:4d} x  {n}")
else:
    want = args or ['seg_colsum', 'fs_transduce', 'fs_chain']
    for s, e, n in sel:
        if any(x in n for x in want):
            print(f"{(s - t0) / 1e3:9.1f} us  {n:28s} {(e - s) / 1e3:8.1f} us")
print('build span', (sel[-1][1] - t0) / 1e3, 'us; kernels', len(sel), 'busy', sum(e - s for s, e, _ in sel) / 1e3)
