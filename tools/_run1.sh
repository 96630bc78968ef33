cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -x -p no:cacheprovider > gpurun_out/pipeline.log 2>&1; echo "pipeline exit $?"; tail -40 gpurun_out/pipeline.log
