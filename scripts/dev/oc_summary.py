import json,sys
d=json.loads([x for x in sys.stdin if x.startswith('{')][-1])
for c in d['other_configs']:
    print(c['config'], round(c['ms_per_forward'],3), round(c['ms_per_forward_eager'],3), [(r['kernel'], r['launches'], round(r['avg_us'],1)) for r in c.get('roofline',[])])
