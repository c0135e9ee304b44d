"""Sampler interface (reference VarAutoEncoder/sampler.py:41-53). Sampling / beam search is inference and out of
scope of the training-step hot path (the reference's samplers do not match its own
decoder signature, §3.4). The trainer only needs an object with update_parameters / process_batch."""


class SamplerBase:
    def update_parameters(self, model):
        self.model = model

    def process_batch(self, batch, output_path, num_classes):
        return None


def get_sampler(name, model_folder, context, checkpoint, args):
    if name not in ("sampling", "beam-search"):
        raise ValueError("unknown sampler " + str(name))
    return SamplerBase()
