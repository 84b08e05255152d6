"""recalgorithm_amd.algorithm.PNN — part of the MI355X-native hot-path mirror."""
