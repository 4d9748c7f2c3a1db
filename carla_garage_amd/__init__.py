"""carla_garage_amd: MI355X-native (gfx950) TransFuser++ hot path -- hand-written HIP kernels behind a C ABI
(include/tfpp.h, csrc/) and the Python boundary module mirroring team_code/model.py::LidarCenterNet."""
__version__ = '0.1.0'
