"""recalgorithm_amd.algorithm.DIN — part of the MI355X-native hot-path mirror."""
