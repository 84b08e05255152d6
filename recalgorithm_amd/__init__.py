"""recalgorithm_amd — part of the MI355X-native hot-path mirror."""
