"""cook_amd — MI355X-native fair-share match engine behind Cook's rank / match / rebalance entry points.

Only the hot path lives here: the HIP sources of libcookmatch.so (cook_amd/csrc), the ctypes binding of its C ABI with the
reference's argument meanings (cook_amd.engine: rank / considerable / match / cycle / rebalance / offers / explain entry points;
cook_amd._abi: the struct layouts), pools over ranks (cook_amd.sharding), the simulator's cycle loop over the engine
(cook_amd.replay) and the synthetic workloads of the benchmark and the tests (cook_amd.synth, cook_amd.workload).
"""
__version__ = "0.1.0"
