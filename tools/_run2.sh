cd /root/repo
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
BENCH_ARGS="--batch 32" bash tools/gpu_profile.sh > gpurun_out/profile_b32.log 2>&1
DB=$(ls gpurun_out/prof/*/*results.db gpurun_out/prof/*results.db 2>/dev/null | head -1)
rm -f gpurun_out/b32_stats.md gpurun_out/b1_stats.md gpurun_out/b1f8_stats.md
python tools/rocpd_stats.py $DB gpurun_out/b32_stats.md > /dev/null
grep '"metric"' gpurun_out/prof_bench.log > gpurun_out/b32_line.json
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
bash tools/gpu_profile.sh > gpurun_out/profile_b1.log 2>&1
DB=$(ls gpurun_out/prof/*/*results.db gpurun_out/prof/*results.db 2>/dev/null | head -1)
python tools/rocpd_stats.py $DB gpurun_out/b1_stats.md > /dev/null
grep '"metric"' gpurun_out/prof_bench.log > gpurun_out/b1_line.json
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
BENCH_ARGS="--fp8" bash tools/gpu_profile.sh > gpurun_out/profile_b1f8.log 2>&1
DB=$(ls gpurun_out/prof/*/*results.db gpurun_out/prof/*results.db 2>/dev/null | head -1)
python tools/rocpd_stats.py $DB gpurun_out/b1f8_stats.md > /dev/null
grep '"metric"' gpurun_out/prof_bench.log > gpurun_out/b1f8_line.json
rm -rf gpurun_out/prof
head -14 gpurun_out/b1_stats.md
