from speedplusbaseline_amd.styleaug import StyleAugmentor  # noqa: F401
