import sys, json
o = json.loads(sys.stdin.read().strip().splitlines()[-1])
ak = o['roofline'].get('all_kernels', {})
rows = sorted(((v['avg_launch_us'] * v['launches_per_step'], k, v) for k, v in ak.items()), reverse=True)
print('ms/step', round(o['ms_per_step'], 4), o['step_ms_percentiles'])
for t, k, v in rows:
    print(f"{t:8.1f} us/step  {v['avg_launch_us']:7.2f} us x {v['launches_per_step']:5.2f}  {k}")
