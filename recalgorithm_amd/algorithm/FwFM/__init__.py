"""recalgorithm_amd.algorithm.FwFM — §8f-3 sibling model on the same hot-path kernels."""
