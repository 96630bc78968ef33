#!/bin/sh
# Entry point with the reference's name and behaviour: runs the two-stage SocioSeg inference pipeline with
# examples/infer/rlvr_megatron.yaml, served by the MI355X-native engine.  Works from any directory; extra arguments are
# passed through.  8 GPUs:
#   torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/start_rlvr_socioseg_pipeline_infer.py \
#       --config_path infer --config_name rlvr_megatron
# Checkpoints: `pretrain` / `model_name_or_path` in the YAML are the reference's hub ids.  They resolve to a directory, or to a snapshot in the local
# HuggingFace cache (never fetched); with neither on disk the run stops with FileNotFoundError -- SR_ALLOW_SYNTHETIC_WEIGHTS=1 runs the same pipeline on
# random weights of the same geometry instead (socioreasoner_amd/checkpoints.py; tools/run_example_small.py does that for the offline demo).
here="$(cd "$(dirname "$0")" && pwd)"
exec python "$here/../start_rlvr_socioseg_pipeline_infer.py" --config_path "$(basename "$here")" --config_name rlvr_megatron "$@"
