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
    "sdplib/mcp124-2.dat-s", "sdplib/mcp124-3.dat-s", "sdplib/mcp124-4.dat-s",
    "sdplib/mcp250-2.dat-s", "sdplib/mcp250-3.dat-s", "sdplib/mcp250-4.dat-s",
    "sdplib/mcp500-2.dat-s", "sdplib/mcp500-3.dat-s", "sdplib/mcp500-4.dat-s",
    "sdplib/gpp100.dat-s", "sdplib/gpp124-1.dat-s", "sdplib/gpp124-2.dat-s", "sdplib/gpp124-3.dat-s", "sdplib/gpp124-4.dat-s",
    "sdplib/gpp250-1.dat-s", "sdplib/gpp250-2.dat-s", "sdplib/gpp250-3.dat-s", "sdplib/gpp250-4.dat-s",
    "sdplib/gpp500-1.dat-s", "sdplib/gpp500-2.dat-s", "sdplib/gpp500-3.dat-s", "sdplib/gpp500-4.dat-s",
    "sdplib/theta5.dat-s", "sdplib/theta6.dat-s",
    "Gset/G60.txt",                              # maxG60 of data/sdplib/README:75 (data/sdplib/maxG60.dat-s holds the same graph)
    "sdplib/theta1.dat-s", "sdplib/theta2.dat-s", "sdplib/theta3.dat-s", "sdplib/theta4.dat-s",
    "bqp_Q_10_1.txt", "bqp_e_10_1.txt", "bqp_Q_20_1.txt", "bqp_e_20_1.txt",
    "bqp_Q_30_1.txt", "bqp_e_30_1.txt", "qs_c_10_1.txt",
    "bqp_Q_60_1.txt", "bqp_e_60_1.txt",          # BASELINE.json configs[2] (n = 1831, m = 1 155 281)
    # data/sdplib/README:104-105.  Listed with the theta family, but NOT unit-trace problems: constraints 1..n are X_ii = 1
    # (n = 801 / 1001 = |V| + 1), the others tie a 3 x 3 all-ones pattern per edge to 1 -> ManiSDP_unitdiag instances
    "sdplib/thetaG11.dat-s", "sdplib/thetaG51.dat-s",
]

# data/sdplib/README:39-51 (gpp), :71-88 (maxG/mcp), :98-105 (theta): optimal objective values, AS PRINTED there (the number
# of printed digits is the accuracy of the pin: tests compare after rounding to the same digits)
PRINTED = {
    "maxG11": "6.291648e+02", "maxG32": "1.567640e+03", "maxG60": "1.522227e+04",
    "mcp100": "2.261574e+02", "mcp124-1": "1.419905e+02", "mcp124-2": "2.698802e+02", "mcp124-3": "4.677501e+02",
    "mcp124-4": "8.644119e+02", "mcp250-1": "3.172643e+02", "mcp250-2": "5.319301e+02", "mcp250-3": "9.811726e+02",
    "mcp250-4": "1.681960e+03", "mcp500-1": "5.981485e+02", "mcp500-2": "1.070057e+03", "mcp500-3": "1.847970e+03",
    "mcp500-4": "3.566738e+03",
    "gpp100": "-4.49435e+01", "gpp124-1": "-7.3431e+00", "gpp124-2": "-4.68623e+01", "gpp124-3": "-1.53014e+02",
    "gpp124-4": "-4.1899e+02", "gpp250-1": "-1.5445e+01", "gpp250-2": "-8.1869e+01", "gpp250-3": "-3.035e+02",
    "gpp250-4": "-7.473e+02", "gpp500-1": "-2.53e+01", "gpp500-2": "-1.5606e+02", "gpp500-3": "-5.1302e+02",
    "gpp500-4": "-1.56702e+03",
    "theta1": "2.300000e+01", "theta2": "3.287917e+01", "theta3": "4.216698e+01", "theta4": "5.032122e+01",
    "theta5": "5.723231e+01", "theta6": "6.347709e+01",
    "thetaG11": "4.000000e+02", "thetaG51": "3.49000e+02",
}
# Not used as a pin: maxG51.  data/sdplib/maxG51.dat-s is Gset G51 (same matrix as data/Gset/G51.txt through either reader);
# the oracle certifies 4006.2555 for it (dinf 5e-12) where README:73 prints 4.003809e+03 -- the file and the printed value
# do not belong together, so the instance pins nothing.  Likewise maxG55 (README:74 prints 9.999210e+03): Gset G55 certifies
# 11039.4604 (dinf 5e-12) and data/sdplib/maxG55.dat-s -- a different graph -- 12869.8667.  maxG60 (Gset G60) does match.
KNOWN = {k: float(v) for k, v in PRINTED.items()}

if __name__ == "__main__":
    for rel in FILES:
        src = os.path.join(REF, rel)
        dst = os.path.join(HERE, os.path.basename(rel) + ".gz")
        with open(src, "rb") as fi, gzip.GzipFile(dst, "wb", mtime=0) as fo:
            shutil.copyfileobj(fi, fo)
        print(dst, os.path.getsize(dst))
    with open(os.path.join(HERE, "known_answers.json"), "w") as fh:
        json.dump(KNOWN, fh, indent=1, sort_keys=True)
    with open(os.path.join(HERE, "known_answers_printed.json"), "w") as fh:
        json.dump(PRINTED, fh, indent=1, sort_keys=True)
