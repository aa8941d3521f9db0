mkdir -p gpurun_out/r05a

T0=$(date +%s); python bench.py > gpurun_out/r05a/bench_default.json 2> gpurun_out/r05a/bench_default.err; echo "bench wall $(( $(date +%s) - T0 )) s"; tail -5 gpurun_out/r05a/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05a/bench_default.json').read().strip().splitlines()[-1])
print(d['value']/1e6, d['roofline']['frac'], d['roofline']['kernel_ms_avg'])
print(json.dumps(d.get('other_configs'), indent=0)[:3000])
c=d['cpu_baseline']; print(c['value'], c['cores'], c['threads'], c['matches_gpu_on_sample'], c['sample_check'])
PY
