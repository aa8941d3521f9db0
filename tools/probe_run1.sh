O=gpurun_out/r05_sweeps; mkdir -p $O
timeout 1500 python tests/parity_sweep.py 24 plain 11000 > $O/parity_plain4.log 2>&1; tail -1 $O/parity_plain4.log
timeout 1500 python tests/parity_sweep.py 24 wide 11100 > $O/parity_wide4.log 2>&1; tail -1 $O/parity_wide4.log
timeout 900 python tests/primer_sweep.py 100 11200 > $O/primer4.log 2>&1; tail -1 $O/primer4.log
timeout 900 python tests/primer_sweep.py 80 11400 after > $O/primer_after4.log 2>&1; tail -1 $O/primer_after4.log
timeout 600 python tests/primer_sweep.py 40 11600 penalty > $O/primer_penalty4.log 2>&1; tail -1 $O/primer_penalty4.log
timeout 900 python tests/length_sweep.py 48 11700 > $O/length4.log 2>&1; tail -1 $O/length4.log
timeout 600 python tests/compat_sweep.py 60 11800 > $O/compat4.log 2>&1; tail -1 $O/compat4.log
timeout 600 python tests/per_pair_sweep.py 40 11900 > $O/per_pair4.log 2>&1; tail -1 $O/per_pair4.log
timeout 600 python tests/textio_sweep.py 100 12000 > $O/textio4.log 2>&1; tail -1 $O/textio4.log
timeout 300 python tests/stress_determinism.py > $O/determinism4.log 2>&1; tail -1 $O/determinism4.log
grep -l "MISMATCH" $O/*4.log
