"""CPU oracle for the VRP-GYM hot path.  TEST INFRASTRUCTURE — NOT PRODUCT CODE.

A plain numpy / torch-CPU restatement of the reference algorithm (environment
bookkeeping E1-E8, policy network N1-N3, decoder D1-D6, rollouts R1-R3; the row
ids are those of SURVEY.md section 8a).  Every function cites the reference
file:line it follows.

Who may import this package: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` — and there only as the checker / the timed
CPU baseline.  The product (``vrp-gym_amd/``) never imports it; the product path
fails loudly when the HIP library is missing instead of falling back to this.

Pinning: the oracle is checked against (a) the reference's own known-answer
tests (tests/test_agent.py:69,84,99,114, tests/test_env.py:48,54-60,
tests/test_graph.py:42) and (b) golden vectors produced by importing the
reference itself in the build container (tools/make_golden.py ->
tests/golden/*.npz).  Greedy rollouts are pinned end to end; sampling is pinned
only against reference outputs generated here (the reference's single sampling
test, tests/test_agent.py:54, fails on the reference itself under this torch).
"""
