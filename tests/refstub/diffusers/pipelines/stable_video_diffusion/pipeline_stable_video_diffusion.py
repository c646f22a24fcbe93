from dataclasses import dataclass


@dataclass
class StableVideoDiffusionPipelineOutput:
    frames: object


def tensor2vid(video, processor, output_type="np"):
    raise NotImplementedError("SVD leaves are not part of the stub")


class StableVideoDiffusionPipeline:
    pass
