import sys, json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
hb=j['config'].get('host_buffers',{})
print(sys.argv[1], j['ms_per_step'], {k:v['ms'] for k,v in hb.items() if isinstance(v,dict) and 'ms' in v})
