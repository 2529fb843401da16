from speedplusbaseline_amd.nets.revgrad import GradientReversalFunction, RevGrad  # noqa: F401
