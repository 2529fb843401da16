from speedplusbaseline_amd.core.trainer import train_single_epoch_krn, train_single_epoch_spn  # noqa: F401
