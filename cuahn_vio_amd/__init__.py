"""cuahn_vio_amd — MI355X-native HomographyNet inference hot path of tudelft/CUAHN-VIO.

Scope: the forward that `cuahn_ros/homography_network` delegates to a traced
TorchScript model, rebuilt as hand-written HIP kernels for gfx950 behind a C ABI (include/hnet.h).
"""
__version__ = "0.1.0"
