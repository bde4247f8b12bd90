"""Reads a rocprofv3 --kernel-trace CSV and reports how much of the busiest stretch had k kernels in flight, and how many
hardware queues carried them: tells whether a multi-stream run is bound by the GPU or by the host that feeds it."""
import csv, glob, sys
files = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
if not rows:
    print('no kernel trace under', sys.argv[1]); sys.exit(0)
k = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?')) for r in rows]
k.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0           # launches before the timed region (uploads)
tail = int(sys.argv[3]) if len(sys.argv) > 3 else 0           # launches after it (decryption)
k = k[skip:len(k) - tail]
t0, t1 = k[0][0], max(e for _, e, _ in k)
ev = []
for b, e, _ in k:
    ev.append((b, 1)); ev.append((e, -1))
ev.sort()
cur = 0; last = t0; hist = {}
for t, d in ev:
    if t > last: hist[cur] = hist.get(cur, 0) + (t - last)
    cur += d; last = t
span = t1 - t0
tot = sum(e - b for b, e, _ in k)
print('%d kernels on %d queues, %.3f s of kernel time in a window of %.3f s (average %.2f in flight)' % (len(k), len(set(q for _, _, q in k)), tot / 1e9, span / 1e9, tot / span))
for n in sorted(hist):
    print('  %2d in flight: %5.1f %%' % (n, 100.0 * hist[n] / span))
