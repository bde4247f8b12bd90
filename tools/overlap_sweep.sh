for args in "--overlap 0" "--overlap 1" "--overlap 0 --chunk 512" "--overlap 1 --chunk 512" "--overlap 1 --chunk 128" "--overlap 0"; do
  python bench.py --steps 10 --warmup 3 --no-mulrelin --no-cpu $args 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); r = j['roofline']
        print('$args: NTT/s %.0f  pipelined %.4f  pass1 %.4f ms  pass2 %.4f ms copy %s' % (j['value'], r['pipelined_ms_per_batch'], r['pass1_ms_per_batch'], r['pass2_ms_per_batch'], r.get('measured_copy_GBs')))
"
done
