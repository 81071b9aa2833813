"""better_fastlio2_amd -- MI355X-native (gfx950, hand-written HIP) measurement path of the LiDAR-inertial
odometry system Yixin-F/better_fastlio2: GPU-resident map index + exact 5-NN + plane fit + point-to-plane
residual/Jacobian reduced on device to the H^T H / H^T h blocks the iterated ESKF consumes.

    csrc/        HIP kernels, the C ABI (include/lio_hip.h) and the host-side ESKF mirror
    capi.py      ctypes binding + Python mirrors of KD_TREE / h_share_model / esekf
    synth.py     seeded synthetic worlds and scans (inputs only)
    replicate.py RCCL/xGMI exchange of accepted map inserts between replicas (one rank per GPU)
    build.py     hipcc build of lib/liblio_hip.so
"""
__version__ = "0.1.0"
