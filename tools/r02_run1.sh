#!/bin/bash
# round-2 GPU call 1: full GPU test suite, bench lines of the new contract, two hardware probes
set -u
mkdir -p gpurun_out
O=gpurun_out/r02a
mkdir -p $O
echo "== ubench" 
timeout -k 10 120 tools/ubench/fmt_load > $O/fmt_load.txt 2>&1; echo "fmt_load rc=$?"; cat $O/fmt_load.txt
timeout -k 10 120 tools/ubench/mfma_f16_numerics > $O/mfma_f16_numerics.txt 2>&1; echo "mfma rc=$?"; cat $O/mfma_f16_numerics.txt
echo "== bench default"
timeout -k 10 300 python3 bench.py > $O/bench_encode4096.json 2> $O/bench_encode4096.err; echo "rc=$?"; tail -c 1500 $O/bench_encode4096.json
for wl in decode4096 encode4096_jpg; do
  timeout -k 10 300 python3 bench.py --workload $wl --steps 100 > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "$wl rc=$?"; tail -c 900 $O/bench_$wl.json
done
echo "== bench --batch (configs[3] at N=1)"
timeout -k 10 400 python3 bench.py --batch --no-cpu --steps 50 > $O/bench_batch_n1.json 2> $O/bench_batch_n1.err; echo "rc=$?"; tail -c 1200 $O/bench_batch_n1.json; tail -3 $O/bench_batch_n1.err
echo "== pytest -m gpu"
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.log
