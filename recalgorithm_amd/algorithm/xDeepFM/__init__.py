"""recalgorithm_amd.algorithm.xDeepFM — part of the MI355X-native hot-path mirror."""
