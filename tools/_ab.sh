cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
MIX="python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --ref-reads 0 --no-e2e --sustain-seconds 0 --no-extra-lanes --mix-steps 5 --k25-parity-pairs 0"
for mix in walk k25; do
for f in 2 8 32; do
  L=""; [ $f != 2 ] && L="--lib danbing-tk_amd/libdbtk_hip_wfr$f.so"
  DBTK_BENCH_DETAIL=/tmp/d_$f.json $MIX $L --only-mix $mix > /tmp/o_$f.txt 2>&1
  python3 -c "
import json
d=json.load(open('/tmp/d_$f.json'))
for n,m in d['mixes'].items():
    print('wfr=$f', n, round(m['ms_per_step'],3), {k:round(v['avg_ms']*v['launches']/5,3) for k,v in m['roofline']['kernels'].items() if 'only' not in k})"
done
done
