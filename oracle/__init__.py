"""TEST INFRASTRUCTURE — CPU restatement ("oracle") of the reference's AGCN / ST-GCN hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package, and only as the checker / reported baseline.  The product (``fusion_gcn_amd``) never imports it.

Parity status: PINNED — every function here is checked in ``tests/test_oracle_golden.py`` against golden
vectors produced by importing the reference itself (``oracle/gen_golden.py``, run in the build container
where ``/root/reference`` is mounted; the vectors live in ``tests/golden/``).
"""
