mkdir -p gpurun_out/r05_sweeps; O=gpurun_out/r05_sweeps
python -m pytest tests/ -x -q -m gpu > $O/tests_all.log 2>&1; tail -3 $O/tests_all.log
timeout 900 python tests/parity_sweep.py 14 plain 9000 > $O/parity_plain.log 2>&1; tail -2 $O/parity_plain.log
timeout 900 python tests/parity_sweep.py 14 wide 9100 > $O/parity_wide.log 2>&1; tail -2 $O/parity_wide.log
timeout 600 python tests/length_sweep.py 24 9200 > $O/length.log 2>&1; tail -2 $O/length.log
timeout 600 python tests/primer_sweep.py 60 9300 > $O/primer.log 2>&1; tail -2 $O/primer.log
timeout 300 python tests/primer_sweep.py 20 9400 after > $O/primer_after.log 2>&1; tail -2 $O/primer_after.log
timeout 300 python tests/primer_sweep.py 20 9500 penalty > $O/primer_penalty.log 2>&1; tail -2 $O/primer_penalty.log
timeout 300 python tests/compat_sweep.py 20 9600 > $O/compat.log 2>&1; tail -2 $O/compat.log
timeout 300 python tests/per_pair_sweep.py 20 9700 > $O/per_pair.log 2>&1; tail -2 $O/per_pair.log
timeout 300 python tests/textio_sweep.py 30 9800 > $O/textio.log 2>&1; tail -2 $O/textio.log
timeout 300 python tests/stress_determinism.py > $O/determinism.log 2>&1; tail -2 $O/determinism.log
