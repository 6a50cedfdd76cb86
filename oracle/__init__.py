"""CPU oracle for the feature-activation hot path.  TEST INFRASTRUCTURE ONLY.

A plain torch-CPU / numpy restatement of the reference's algorithm for the path named in
``BASELINE.json:north_star``.  Every function cites the reference
``file:line`` it follows.  It is *pinned*: ``tests/test_oracle_golden.py`` checks it against
the fixtures in ``tests/golden/*.npz``, which were produced by importing and running the
reference's own ``SparseTranscoder``, ``ZSAE`` and ``CircuitLens`` in the build container
(``tests/golden/make_golden.py``).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this package, and only as the checker / the timed CPU baseline.  Nothing under
``transcoder-sae-circuits_amd/`` imports it; the product path raises when the HIP library
is missing instead of falling back to anything in here.
"""
