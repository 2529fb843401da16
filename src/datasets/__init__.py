"""src.datasets of the reference: build.make_dataloader and transforms.build_transforms, GPU-batched here
(speedplusbaseline_amd.datasets / speedplusbaseline_amd.transforms)."""
