#!/bin/bash
# Drop-in for the reference's examples/infer/infer.sh: same entry point, same flags.
# Multi-GPU: torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/start_rlvr_socioseg_pipeline_infer.py ...
set +x
CONFIG_PATH=$(basename $(dirname $0))
python examples/start_rlvr_socioseg_pipeline_infer.py --config_path $CONFIG_PATH  --config_name rlvr_megatron
