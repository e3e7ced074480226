"""ORACLE -- test infrastructure only.

CPU restatement of the reference algorithm (Manopt RTR/tCG + the three ManiSDP
entry points).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import anything from here; the product package
``manisdp-matlab_amd`` never does.
"""
