cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
t0=$(date +%s)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4/bench.json 2> gpurun_out/r4/bench.err
echo "rc $? seconds $(( $(date +%s) - t0 ))"
tail -c 600 gpurun_out/r4/bench.err
python3 -c "
import json
d=json.loads(open('gpurun_out/r4/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['cpu_baseline']['value'])
s=d.get('sparse_engine',{})
print('sparse', s.get('value'), s.get('cpu_baseline',{}).get('value'))
print('scale', s.get('scale',{}).get('lu',{}).get('value'), 'replicas', {k:round(v['value']) for k,v in s.get('replicas',{}).get('replicas',{}).items()})
"
