"""Aggregates the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes)
into profiles/<tag>_pmc_traffic.json: per kernel symbol, KB per launch and corrected HBM bytes per launch.

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of the bytes of coalesced streaming reads;
checked here on kernels of known traffic in this very run (bn_apply_kernel: float4 read == float4 write, FETCH = 0.50 x WRITE;
bn_bwd_apply_kernel: two 4-B/lane read streams per written stream, FETCH = 1.00 x WRITE), so
hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024."""
import collections, csv, glob, json, re, sys

def load(d):
  f = glob.glob(d + '/*/*counter_collection.csv')[0]
  agg = collections.defaultdict(lambda: [0, 0.0])
  for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    n = re.sub(r'^void ', '', n).replace('ms::', '')
    n = re.sub(r'\(.*$', '', n).replace(', ', ',').replace('false', '0').replace('true', '1')
    agg[n][0] += 1
    agg[n][1] += float(r['Counter_Value'])
  return agg

fetch, write = load(sys.argv[1]), load(sys.argv[2])
out = {}
for n, (c, v) in fetch.items():
  w = write.get(n, [c, 0.0])
  f_kb, w_kb = v / c, w[1] / max(1, w[0])
  out[n] = dict(launches=c, fetch_size_kb_per_launch=round(f_kb, 1), write_size_kb_per_launch=round(w_kb, 1),
                hbm_bytes_per_launch=int((2 * f_kb + w_kb) * 1024))
json.dump(dict(command='rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python3 bench.py --steps 4 --warmup 2 --no-graphs '
                       '--no-cpu-baseline --no-kernel-timing (two separate passes)',
               correction='hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts half of streaming reads)',
               kernels=dict(sorted(out.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches']))),
          open(sys.argv[3], 'w'), indent=1)
print('wrote', sys.argv[3], len(out), 'kernels')
