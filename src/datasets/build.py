from speedplusbaseline_amd.datasets import make_dataloader  # noqa: F401  (reference: src/datasets/build.py:48-66)
