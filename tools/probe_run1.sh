mkdir -p gpurun_out/r05a
python -m pytest tests/test_pipe_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 1200 python tools/pipe_file_rate.py --pairs 64000000 --reps 2 --out gpurun_out/r05a/pipe_file_64m.json 2>&1 | tail -1 | cut -c1-1500
timeout 1200 python tools/pipe_file_rate.py --pairs 64000000 --reps 2 --sink /dev/null --out gpurun_out/r05a/pipe_null_64m.json 2>&1 | tail -1 | cut -c1-1500
PAMD_PIPE_READERS=5 timeout 1200 python tools/pipe_file_rate.py --pairs 64000000 --reps 2 --sink /dev/null --out gpurun_out/r05a/pipe_null_64m_r5.json 2>&1 | tail -1 | cut -c1-1500
