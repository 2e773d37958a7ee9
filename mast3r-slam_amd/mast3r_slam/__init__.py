"""mast3r_slam - MI355X-native hot path of MASt3R-SLAM behind the reference operator API.

Host side: Python on PyTorch-ROCm (device memory, streams, torch.distributed);
device side: hand-written HIP for gfx950 in libm3slam_hip.so (C ABI: include/m3slam.h).
Module split mirrors /root/reference/src/mlx_mast3r_slam: kernels (level-1 array ops),
matching, tracker, mast3r_utils (level-3 operator API), config, frame.
"""
__version__ = "0.1.0"
