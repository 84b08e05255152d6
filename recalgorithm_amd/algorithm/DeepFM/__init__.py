"""recalgorithm_amd.algorithm.DeepFM — part of the MI355X-native hot-path mirror."""
