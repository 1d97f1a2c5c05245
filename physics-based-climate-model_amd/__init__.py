"""MI355X-native hot path of the unet_convlstm_attention climate emulator.

Layout: ``csrc/`` HIP kernels + C-ABI (``include/climate_hip.h``), ``_lib.py`` ctypes binding, ``ops.py`` tensor-level
wrappers (torch is used for device memory, streams and torch.distributed only), ``model.py`` the drop-in
``AttUNetConvLSTM`` / ``get_model`` mirror of the reference's seam, ``lightning_module.py`` the LightningModule mirror,
``config.py`` the Hydra-compatible YAML loader, ``ddp.py`` data-parallel gradient exchange over RCCL.
"""
__version__ = "0.1.0"
