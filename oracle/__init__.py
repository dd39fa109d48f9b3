"""CPU oracle for the MSM hot path -- TEST INFRASTRUCTURE ONLY.

This package is a restatement, in plain Python integers (and, next to it, plain C in
``msm_oracle.c``), of the reference's big-integer definition of the path
(``/root/reference/src/bigint/*.ts``).  It is the checker, never the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import it.  Nothing under ``msm_zprize_amd/`` imports it.

Parity status: PINNED BY THE REFERENCE'S OWN FIXTURES ONLY.  The reference is TypeScript that
needs ``tsc`` + ``wasmati`` (absent, no network) and a node >= 18 (node 12 here), so it cannot be
run in this container.  The oracle is therefore checked against every
fixed vector the reference's tests hold for this path (``tests/test_oracle.py``):
the BLS12-377 and ed-on-bls12-377 known-answer points and the 2P+(q-1)P=P identity
(scripts/zprize23/submission-test-bls377.ts:6-25, submission-test.ts:5-20), the four
generators, and the beta/lambda endomorphism relations (concrete/bls12-377.params.ts:49-63).
"""
