timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_primers_after_gpu.py -x -q -k "exp_collapses or primer or config2 or golden" 2>&1 | tail -3
python bench.py --no-cpu --no-e2e --steps 2 --config 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value']/1e6, d['roofline']['kernel_ms_avg'])"
cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/pk -o t -- python3 /root/repo/bench.py --no-cpu --no-e2e --steps 2 --config 2 > /dev/null 2>&1; cd /root/repo
python3 profiles/summarize_rocprof.py --like "%pamd_%" $(find gpurun_out/pk -name "*_results.db" | head -1) | head -5
