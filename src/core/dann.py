from speedplusbaseline_amd.core.dann import train_dann_single_epoch_krn  # noqa: F401
