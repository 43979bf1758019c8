"""gbp_poplar_amd — MI355X-native Gaussian-Belief-Propagation bundle adjustment (hot path of joeaortiz/gbp-poplar).

Layout: csrc/ (HIP kernels + C-ABI + C++ host/CLI), _cabi.py (ctypes structs of include/gbp_mi355x.h),
hostlib.py / engine.py (bindings of the C-ABI), driver.py (the reference's main() flow), distributed.py
(landmark sharding over torch.distributed).  Nothing here imports the CPU oracle under oracle/.
"""
__version__ = "0.1.0"
