"""cook_amd — MI355X-native fair-share match engine behind Cook's rank / match / rebalance entry points.

Only the hot path lives here: the host-side mirror of the reference interface (cook_amd.scheduler,
cook_amd.rebalancer), the ctypes binding of the C ABI (cook_amd.engine) and the HIP sources (cook_amd/csrc).
"""
__version__ = "0.1.0"
