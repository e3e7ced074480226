"""Option sweep for the G81 solve (device RTR + device escape)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manisdp_matlab_amd import problems, solvers
C = problems.maxcut_cost_matrix(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "G81.txt.gz"))
sets = {
    "p0=40 (example_maxcut.m:32)": {"p0": 40},
    "p0=40, AL_maxiter=60": {"p0": 40, "AL_maxiter": 60},
    "p0=2, AL_maxiter=60": {"p0": 2, "AL_maxiter": 60},
    "mc settings (example/settings.txt)": {"p0": 2, "theta": 1e-2, "delta": 10, "alpha": 0.1, "AL_maxiter": 60},
    "p0=32 alpha=0.1": {"p0": 32, "alpha": 0.1, "AL_maxiter": 60},
}
only = sys.argv[1:] 
for name, o in sets.items():
    if only and not any(k in name for k in only): continue
    t = time.time()
    Y, obj, data = solvers.ManiSDP_onlyunitdiag(C, dict(o), verbose=False)
    print(json.dumps({"set": name, "obj": obj, "dinf": data["dinf"], "status": data["status"], "iters": data["iters"], "time": round(time.time() - t, 2),
                      "hessvecs": data["hessvecs"], "rtr_s": round(data["rtr_seconds"], 2), "eig_s": round(data["eig_seconds"], 2), "p": data["p"],
                      "dinf_trace": ["%.1e" % l[2] for l in data["log"]][-12:]}), flush=True)
