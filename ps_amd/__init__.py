"""ps_amd -- MI355X-native hot path of the wudikua/ps parameter-server trainer.

The package is a thin host mirror (api.py) over the C ABI of
include/ps_native.h, implemented by hand-written HIP kernels for gfx950
(csrc/).  Importing the package does not load the shared object; the first
use does, and fails loudly if it was not built (`python -m ps_amd.build`).
"""
from . import native  # noqa: F401
from .api import (AUC, AdamUpdater, Batch, DataSet, DeviceBatch, DNN, FtrlUpdater, KVStore, LibsvmParser, Mod,  # noqa: F401
                  SimpleUpdater, Trainer, Updater, WideDeepNN, java_string_hash)

__all__ = ["AUC", "AdamUpdater", "Batch", "DataSet", "DeviceBatch", "DNN", "LibsvmParser", "FtrlUpdater", "KVStore", "Mod", "SimpleUpdater", "Trainer",
           "Updater", "WideDeepNN", "java_string_hash", "native"]
