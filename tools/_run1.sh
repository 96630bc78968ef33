cd /root/repo
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 32 --continuous > gpurun_out/b32c.log 2>&1; grep -o '"value": [0-9.]*, "unit": "tiles/s"' gpurun_out/b32c.log || tail -20 gpurun_out/b32c.log
