from speedplusbaseline_amd.core.inference import valid_krn, valid_spn  # noqa: F401
