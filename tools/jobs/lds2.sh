set -x
( while sleep 45; do date >> gpurun_out/heartbeat.txt; done ) &
HB=$!
timeout -k 10 600 python -u -m pytest tests/test_gpu_ordered.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_lds2_tests.log 2>&1 || { kill $HB; exit 1; }
timeout -k 10 800 python -u tools/ordered_stress.py --rays 2000000 --rounds 4 --scenes grid,merged --grid 100 > gpurun_out/r3_lds2_stress.log 2>&1 || { kill $HB; exit 1; }
kill $HB
