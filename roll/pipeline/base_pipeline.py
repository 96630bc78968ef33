"""``BasePipeline`` (reference: roll/pipeline/base_pipeline.py:21-91) reduced to what inference needs."""


class BasePipeline:
    def __init__(self, pipeline_config):
        self.pipeline_config = pipeline_config

    def run(self):
        raise NotImplementedError

    def model_update(self, global_step):
        return {}      # weights are loaded once by the engine; no trainer -> engine sync on the infer path
