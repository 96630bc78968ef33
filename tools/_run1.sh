cd /root/repo
for pb in 1 32; do PB=$pb timeout 200 python tools/bench_gemv_f8.py 2>&1 | grep "f8"; PB=$pb timeout 200 python tools/bench_gemv.py 2>&1 | grep -v lm_head | grep "TB/s"; done
