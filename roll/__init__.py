"""API surface of the reference's inference path, kept so that ``examples/infer/infer.sh`` is a drop-in.

Only the names the infer path touches exist here (SURVEY.md section 8(B) "API surface to keep verbatim"); the
implementation behind them is ``socioreasoner_amd`` (MI355X-native).  Ray, vLLM, Megatron, DeepSpeed, the RL
training pipelines and everything else of the reference's control plane are out of scope.
"""
