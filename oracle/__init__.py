"""CPU oracle for the OAK Gram / SGPR-ELBO hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / CPU comparator.
The shipped package (``orthogonal-additive-gaussian-processes_amd/oak``) never
imports this module and fails loudly when the HIP library is missing.

Parity pinning status: the reference is pure Python on
TensorFlow/GPflow/TFP, none of which is installed or installable here, so the
reference itself cannot be executed to generate vectors.  The oracle is pinned
against (a) every exact property / closed-form answer the reference's own test
suite holds for this path (re-expressed in tests/test_oracle_*.py), and (b) a
50-digit mpmath definitional restatement (dense Titsias bound, dense GP
posterior, quadrature of the Sobol integrals).  Against reference-EXECUTED
outputs the ELBO scalar / predictive variance are "parity unpinned".
"""
