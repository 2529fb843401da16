from speedplusbaseline_amd.nets.park2019 import ConvDw, RouterV2, RouterV3, KeypointRegressionNet  # noqa: F401
