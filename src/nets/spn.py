from speedplusbaseline_amd.nets.spn import SpacecraftPoseNet, softmax_cross_entropy_with_logits  # noqa: F401
