"""Full ManiSDP_onlyunitdiag solve of Gset G81 on the GPU (device RTR + device escape)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import problems, solvers
C = problems.maxcut_cost_matrix(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "G81.txt.gz"))
opts = {"p0": int(sys.argv[1]) if len(sys.argv) > 1 else 2}
if len(sys.argv) > 2: opts["AL_maxiter"] = int(sys.argv[2])
if len(sys.argv) > 3: opts["eig"] = sys.argv[3]
t = time.time()
Y, obj, data = solvers.ManiSDP_onlyunitdiag(C, opts, verbose=True)
print(json.dumps({"obj": obj, "dinf": data["dinf"], "status": data["status"], "time": time.time() - t, "hessvecs": data["hessvecs"],
                  "rtr_s": data["rtr_seconds"], "eig_s": data["eig_seconds"], "p": data["p"], "iters": data["iters"]}))
