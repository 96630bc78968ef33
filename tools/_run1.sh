cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "tiny or decode_graph or decode_step" -p no:cacheprovider > gpurun_out/tiny.log 2>&1; echo "tiny exit $?"; tail -5 gpurun_out/tiny.log
timeout 600 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -p no:cacheprovider > gpurun_out/pipeline.log 2>&1; echo "pipeline exit $?"; tail -15 gpurun_out/pipeline.log
timeout 300 python tools/probe_prefetch.py > gpurun_out/probe_prefetch.log 2>&1; echo "probe exit $?"; cat gpurun_out/probe_prefetch.log
