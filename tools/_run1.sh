cd /root/repo
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --fp8 > gpurun_out/b1f8.log 2>&1; grep -o '"value": [0-9.]*, "unit": "tiles/s"\|"decode_step_ms": [0-9.]*\|"achieved": [0-9.]*\|"avg_launch_us": [0-9.]*\|"frac": [0-9.]*' gpurun_out/b1f8.log || tail -20 gpurun_out/b1f8.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --fp8 --batch 32 > gpurun_out/b32f8.log 2>&1; grep -o '"value": [0-9.]*, "unit": "tiles/s"\|"decode_step_ms": [0-9.]*' gpurun_out/b32f8.log || tail -20 gpurun_out/b32f8.log
