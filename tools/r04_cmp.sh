for mode in 1 2; do
  OVR_HIP_LAYOUTS=$mode python bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read()); s=b['with_empty_space_skipping']
print('layouts mode $mode', round(b['ms_per_step'],3), 'skip', round(s['ms_per_step'],3), s['layout'], s['pipeline'], s['tuning'], s['phase_ms_last_frame'], {k:(round(v['ms_per_step'],3), v['layout'], v['pipeline'], v['extra_warmup'], {a:round(x,2) for a,x in v['phase_ms'].items()}) for k,v in b['roofline']['views'].items()})"
done
