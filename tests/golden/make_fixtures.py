"""Regenerate the data fixtures under tests/golden/ from the reference tree.

Run in the build container (where /root/reference exists):
    python tests/golden/make_fixtures.py
Fixtures are DATA only: problem instances that the reference ships under data/
(Gset graphs, SDPLIB .dat-s files, BQP / quartic coefficient files) gzip-compressed,
plus known_answers.json with the optimal values of data/sdplib/README:39-51,71-88,98-105.
The GPU box has no /root/reference, so the tests read these copies.
"""
import gzip
import json
import os
import shutil

REF = "/root/reference/data"
HERE = os.path.dirname(os.path.abspath(__file__))

FILES = [
    "Gset/G1.txt", "Gset/G11.txt", "Gset/G81.txt", "Gset/G32.txt",
    "sdplib/mcp100.dat-s", "sdplib/mcp124-1.dat-s", "sdplib/mcp250-1.dat-s", "sdplib/mcp500-1.dat-s",
    "sdplib/gpp100.dat-s", "sdplib/gpp124-1.dat-s",
    "sdplib/theta1.dat-s", "sdplib/theta2.dat-s",
    "bqp_Q_10_1.txt", "bqp_e_10_1.txt", "bqp_Q_20_1.txt", "bqp_e_20_1.txt",
    "bqp_Q_30_1.txt", "bqp_e_30_1.txt", "qs_c_10_1.txt",
    "bqp_Q_60_1.txt", "bqp_e_60_1.txt",          # BASELINE.json configs[2] (n = 1831, m = 1 155 281)
]

# data/sdplib/README:39-51 (gpp), :71-88 (maxG/mcp), :98-105 (theta): optimal objective values
KNOWN = {
    "maxG11": 6.291648e+02, "maxG32": 1.567640e+03,
    "mcp100": 2.261574e+02, "mcp124-1": 1.419905e+02, "mcp250-1": 3.172643e+02, "mcp500-1": 5.981485e+02,
    "gpp100": -4.49435e+01, "gpp124-1": -7.3431e+00,
    "theta1": 2.300000e+01, "theta2": 3.287917e+01,
}

if __name__ == "__main__":
    for rel in FILES:
        src = os.path.join(REF, rel)
        dst = os.path.join(HERE, os.path.basename(rel) + ".gz")
        with open(src, "rb") as fi, gzip.GzipFile(dst, "wb", mtime=0) as fo:
            shutil.copyfileobj(fi, fo)
        print(dst, os.path.getsize(dst))
    with open(os.path.join(HERE, "known_answers.json"), "w") as fh:
        json.dump(KNOWN, fh, indent=1, sort_keys=True)
