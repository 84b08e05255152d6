"""oracle/ — TEST INFRASTRUCTURE ONLY.

CPU restatement of the tangxyw/RecAlgorithm CTR hot path, used
as the parity checker for the HIP kernels.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
package.  Nothing under `recalgorithm_amd/` imports it; the product path fails
loudly when the HIP library is missing rather than falling back to this code.

Pinning status:
  * The reference ships no tests, golden vectors or fixtures, and its only
    arithmetic dependency (TensorFlow 1.14) is un-vendored and not installable
    here, so TF's *kernels* cannot be executed: primitive-level parity is
    UNPINNED ("parity unpinned" per SURVEY.md §8c).
  * The *composition* (which primitive, in which order, with which quirk) IS
    pinned: `oracle/gen_golden.py` imports the reference's own, unmodified
    layer/model sources from /root/reference against `oracle/tf1_shim`
    (a numpy implementation of the ~60 documented TF-1.14 primitives those
    files call) and writes `tests/golden/*.npz`.  `ref_ops.py` (an independent
    torch restatement) must reproduce those vectors.
"""
