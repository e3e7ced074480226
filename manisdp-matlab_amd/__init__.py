"""manisdp-matlab_amd: MI355X-native hot path for ManiSDP's three primal entry points.

Only what the hot path needs lives here:

* ``csrc/``      hand-written HIP kernels (gfx950) + the C-ABI library ``libmanisdp_hip.so``
* ``_lib``       ctypes binding of ``include/manisdp_hip.h`` (fails loudly when the library is missing)
* ``solvers``    host-side mirrors of ``ManiSDP_onlyunitdiag / ManiSDP_unitdiag / ManiSDP_unittrace``
                 (same options, same outputs; the augmented-Lagrangian loop stays on the host)
* ``problems``   instance readers/generators (Gset, SDPA, bqpmom, qsmom, theta)
* ``sharding``   row partition used by the multi-GPU path
* ``matlab/``    the MATLAB drop-ins and the MEX shim over the same C-ABI
"""
__version__ = "0.1.0"
