cd /root/repo
for v in 0 256 128; do echo "== SR_GEMM3=$v"; SR_GEMM3=$v timeout 300 python tools/bench_gemm.py 2>&1 | grep -E "B32|square"; done
echo "== correctness"; timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "gemm" -p no:cacheprovider 2>&1 | tail -3
SR_GEMM3=256 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "gemm" -p no:cacheprovider 2>&1 | tail -3
SR_GEMM3=128 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "gemm" -p no:cacheprovider 2>&1 | tail -3
