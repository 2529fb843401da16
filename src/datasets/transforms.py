from speedplusbaseline_amd.transforms import build_transforms, GpuBatchTransform  # noqa: F401  (reference: src/datasets/transforms.py:217-244)
