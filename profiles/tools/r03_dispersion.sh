# step time against the dispersion of the coverage: configs[2]'s geometry, negative-binomial coverage with variance = od x mean
for od in 1.5 3 10 30 100; do
  echo "== overdispersion $od"
  HF_HOST_TRACE=2 python bench.py --config 6 --overdispersion $od --no-cpu-baseline 2> /tmp/disp.err | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('ms_per_step', round(d['ms_per_step'],4), 'windows/s %.3g' % d['value'], {k: round(v*1000,1) for k,v in d['roofline']['kernel_ms_all'].items()})"
  grep -m1 "emission keys" /tmp/disp.err
done
