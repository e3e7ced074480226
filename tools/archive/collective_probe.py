"""Stream time of the collective calls a row-sharded tCG trip issues, on a one-member RCCL communicator (what one GPU can
measure: launch + protocol overhead of the calls, no link traffic), and the trip time with both trip kinds.
    python tools/collective_probe.py [p]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manisdp_matlab_amd import _lib, problems  # noqa: E402

p = int(sys.argv[1]) if len(sys.argv) > 1 else 32
_lib.load()
C = problems.maxcut_cost_matrix(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "G81.txt.gz"))
n = C.shape[0]
rng = np.random.default_rng(0)
Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
out = {"n": n, "p": p}
for halo in (0, 1):
    for trip1 in (1, 0):
        h = _lib.Handle.onlyunitdiag(C, pcap=p)
        h.comm_init(1, 0, _lib.Handle.comm_unique_id())
        h.set_option("halo_exchange", halo)
        h.set_option("trip1", trip1)
        h.set_point(Y)
        h.cost()
        key = "halo%d_trip1_%d" % (halo, trip1)
        out[key] = {"trip_us": 1e3 * h.bench_tcg_trip(256)}
        if trip1:
            names = ["exchange", "allreduce_1", "exchange_with_sums", "allreduce_3", "exchange_then_sums_ungrouped"]
            out[key]["collective_us"] = {nm: h.time_collective(i, 200) for i, nm in enumerate(names)}
        h.close()
print(json.dumps(out, indent=1))
