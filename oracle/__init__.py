"""CPU oracle for the MMFN training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``mmfn_amd/`` may import this package;
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` use it, and there only as the checker / reported baseline, never as
the product path.

Parity pinning: the reference (Kin-Zhang/mmfn) ships no tests or golden vectors
for this path.  The oracle is therefore pinned against
outputs of the reference itself, produced by ``oracle/make_golden.py`` which
imports ``/root/reference/team_code/mmfn_utils`` read-only in the authoring
container and writes the small fixtures committed under ``tests/golden/``.
"""
