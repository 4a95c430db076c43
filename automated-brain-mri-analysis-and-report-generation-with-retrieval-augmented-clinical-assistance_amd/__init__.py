"""MI355X-native nnU-Net (BraTS) sliding-window predictor - host side.

The package holds only what the one hot path needs: the HIP kernels and the
C ABI under ``csrc/`` and the Python mirror of the reference's interfaces for that path.
"""
from . import _build, _isa_gate, _lib, synthetic, network, ops, predictor, preprocessing, nifti, checkpoint, driver, parallel, evaluate, retrieval  # noqa: F401
from .network import UNet, topology_from_state_dict  # noqa: F401
from .predictor import predict_folds, predict_preprocessed_data_return_seg_and_softmax  # noqa: F401

__all__ = ["UNet", "topology_from_state_dict", "synthetic", "network", "ops", "predictor", "predict_folds",
           "predict_preprocessed_data_return_seg_and_softmax"]
